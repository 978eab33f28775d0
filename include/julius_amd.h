/*
 * julius_amd.h -- C ABI of the MI355X (gfx950) engine for Julius' frame-
 * synchronous first pass: acoustic scoring (GMM / tied-mixture / DNN) and
 * first-pass token passing.  Plain pointers and sizes only; no torch, no
 * reference types.  The reference-side binding (a C shim compiled inside the
 * Julius tree that flattens HTK_HMM_INFO / DNNData / WCHMM_INFO and exports the
 * reference's own symbols) is julius_amd/shim/ and is described in
 * INTEGRATION.md.  All paths cited below are relative to the reference root.
 *
 * Conventions
 *   - every entry point returns 0 on success and a negative JAMD_E* code on
 *     failure; jamd_last_error() returns a message for the calling thread.
 *     (The reference's boolean/LOG_ZERO error returns are produced by the shim
 *     from these codes.)
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device
 *     memory on the engine's device; `stream` is a hipStream_t passed as void*
 *     (NULL = the engine's own stream).
 *   - scores are fp32 log10 likelihoods exactly as LOGPROB in
 *     libsent/include/sent/stddefs.h:194; LOG_ZERO = -1000000 (:171).
 *   - an engine object is not re-entrant (neither is HMMWork,
 *     libsent/include/sent/hmm_calc.h:90-111); use one per thread/stream.
 */
#ifndef JULIUS_AMD_H
#define JULIUS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JAMD_ABI_VERSION 4

#define JAMD_OK        0
#define JAMD_EINVAL   -1   /* bad argument / unsupported model feature        */
#define JAMD_ENODEV   -2   /* no usable gfx950 device / HIP runtime failure   */
#define JAMD_ENOMEM   -3
#define JAMD_ELAUNCH  -4   /* kernel launch or execution failed               */
#define JAMD_ESTATE   -5   /* call sequence violation                         */

#define JAMD_LOG_ZERO (-1000000.0f)

/* Gaussian pruning selector (GPRUNE_SEL_*, libsent/include/sent/hmm_calc.h:38-45).  All four are served for
 * every model.  heu and beam prune with thresholds taken from the N best Gaussians of the PREVIOUS frame's
 * codebook cache (last_id), which exists only for tied-mixture codebooks (calc_tied_mix.c:203-215).  For plain
 * mixture states calc_mix() passes last_id == NULL (calc_mix.c:63), where both functions ARE safe pruning
 * (gprune_heu.c:337-350, gprune_beam.c:337-350).  For tied-mixture codebooks the previous frame's cache depends, in
 * the reference, on which frames its lazy search happened to score; the device scores every state of every frame,
 * so its results are the reference's under eager scoring (outprob_set_batch_computation(), outprob.c:230-242) --
 * bit for bit, cache contents included -- and an utterance's first frame takes the no-history branch. */
#define JAMD_GPRUNE_NONE 0  /* gprune_none()  libsent/src/phmm/gprune_none.c:133 */
#define JAMD_GPRUNE_SAFE 1  /* gprune_safe()  libsent/src/phmm/gprune_safe.c:160 */
#define JAMD_GPRUNE_HEU  2  /* gprune_heu()   gprune_heu.c:295  (tied-mixture codebooks: parity under eager scoring) */
#define JAMD_GPRUNE_BEAM 3  /* gprune_beam()  gprune_beam.c:291 (ditto) */

/* pseudo-phone set reduction (hmminfo->cdset_method, htk_hmm.h:388) */
#define JAMD_IWCD_MAX   0   /* outprob_cd_max   libsent/src/phmm/outprob.c:332 */
#define JAMD_IWCD_AVG   1   /* outprob_cd_avg   outprob.c:356 */
#define JAMD_IWCD_NBEST 2   /* outprob_cd_nbest outprob.c:287 */

typedef struct jamd_engine jamd_engine;
typedef struct jamd_gmm    jamd_gmm;
typedef struct jamd_cdset  jamd_cdset;
typedef struct jamd_dnn    jamd_dnn;

/* ------------------------------------------------------------------ engine */
int         jamd_abi_version(void);
const char *jamd_last_error(void);
int         jamd_device_count(void);

/* Create an engine on HIP device `device`.  Builds the addlog table of
 * make_log_tbl() (libsent/src/phmm/addlog.c:42) and the logistic table of
 * logistic_table_build() (libsent/src/phmm/calc_dnn.c:349) on the host with the
 * same libm expressions and uploads them.  Replaces the table part of
 * outprob_init() (libsent/src/phmm/outprob_init.c:161). */
int  jamd_engine_create(int device, jamd_engine **out);
void jamd_engine_destroy(jamd_engine *e);
int  jamd_engine_device(const jamd_engine *e);
int  jamd_engine_sync(jamd_engine *e);
/* Device allocation helpers for callers without their own HIP runtime. */
int  jamd_malloc(jamd_engine *e, size_t bytes, void **dev);
int  jamd_free(jamd_engine *e, void *dev);
/* Page-locked host memory: copies to and from it run at the PCIe rate instead of the pageable-memory rate
 * (about 3x); the bindings keep the score rows they hand to Julius' outprob cache in such buffers. */
int  jamd_host_alloc(jamd_engine *e, size_t bytes, void **host);
int  jamd_host_free(jamd_engine *e, void *host);
int  jamd_memcpy_h2d(jamd_engine *e, void *dev, const void *host, size_t bytes);
int  jamd_memcpy_d2h(jamd_engine *e, void *host, const void *dev, size_t bytes);
/* Streams for callers without their own HIP runtime (the `stream` arguments below take any hipStream_t): a batch
 * driver scores input k+1 on one stream while the first pass of input k runs on another -- the first pass keeps
 * one workgroup per utterance busy, the scoring kernels fill whatever CUs that leaves (host/jamd_batch.c).
 * jamd_stream_wait(): everything submitted to `waiter` after the call starts only when everything submitted to
 * `signaler` before the call is done (an event record + wait; NULL = the engine's own stream).
 * Two things such a driver has to know: (1) the first pass must be ON the device before the next input's scoring is
 * queued (jamd_beam_wait_started() below), or the scoring workgroups take the LDS it is waiting for; (2) the ROCm
 * runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two streams
 * on one queue run one after the other -- a process that creates many streams raises that limit before its first HIP
 * call (jamd_batch and bench.py set 16). */
int  jamd_stream_create(jamd_engine *e, void **stream);
int  jamd_stream_destroy(jamd_engine *e, void *stream);
int  jamd_stream_wait(jamd_engine *e, void *waiter, void *signaler);
int  jamd_stream_sync(jamd_engine *e, void *stream);
int  jamd_memcpy_h2d_async(jamd_engine *e, void *dev, const void *host, size_t bytes, void *stream);

/* --------------------------------------------------------------- GMM model */
/* Flattened HTK_HMM_INFO (libsent/include/sent/htk_hmm.h:330-420) after
 * outprob_init() has inverted the variances (outprob_init.c:74-79):
 *   densities   mean[ndens][veclen], ivar[ndens][veclen], gconst[ndens]
 *               (HTK_HMM_Dens, htk_hmm.h:120-131; gconst as computed by
 *               libsent/src/hmminfo/rdhmmdef_dens.c:39-48)
 *   states      st_off[nstate+1] into the entry arrays, in HTK_HMM_State.id order
 *   entries     ent_dens[nentry] (density index, -1 for a NULL density),
 *               ent_logw[nentry] (bweight = ln w, rdhmmdef_mpdf.c:189)
 *   st_book     [nstate] codebook id of a tied-mixture state (its entries list
 *               the GCODEBOOK's densities in codebook order, htk_hmm.h:196-201),
 *               -1 for a plain state; may be NULL when nbook == 0.
 * Single-stream models only (nstream must be 1; multi-stream -> JAMD_EINVAL). */
typedef struct {
  int nstate, veclen, ndens, nentry, nbook, nstream;
  const float *mean, *ivar, *gconst;
  const int   *st_off;
  const int   *ent_dens;
  const float *ent_logw;
  const int   *st_book;
} jamd_gmm_desc;

/* Replaces outprob_init()'s selection of compute_gaussset/calc_outprob
 * (outprob_init.c:99-147): gprune = JAMD_GPRUNE_*, gprune_num = -tmix /
 * OP_gprune_num (ignored for NONE).  Uploads the model in the engine's device
 * layout (DESIGN.md "HBM layout"). */
int  jamd_gmm_create(jamd_engine *e, const jamd_gmm_desc *d, int gprune, int gprune_num,
                     jamd_gmm **out);
/* The same from a "JAMDGMM1" file: the flattened model as jamd_gmm_save()
 * (julius_amd/shim/jamd_flatten.c) wrote it from the HTK_HMM_INFO that Julius' own
 * hmmdefs / binhmm reader (libsent/src/hmminfo/rdhmmdef.c, read_binhmm.c) built.
 * For workers that never link Julius. */
int  jamd_gmm_load(jamd_engine *e, const char *path, int gprune, int gprune_num, jamd_gmm **out);
void jamd_gmm_destroy(jamd_gmm *g);
int  jamd_gmm_nstate(const jamd_gmm *g);
int  jamd_gmm_veclen(const jamd_gmm *g);

/* Score every state for frames 0..T-1: out[t][s] = log10 b_s(o_t), the values
 * outprob_state() (libsent/src/phmm/outprob.c:184) would cache in
 * outprob_cache[t][s] via calc_mix() (calc_mix.c:41) / calc_tied_mix()
 * (calc_tied_mix.c:162) with the batch loop outprob.c:230-242.
 * frames is [T][veclen] row-major, out is [T][nstate] row-major (the
 * outprob_cache layout, outprob.c:127-129). */
int  jamd_gmm_outprob_dev(jamd_gmm *g, const float *dev_frames, int T, float *dev_out,
                          void *stream);
/* The same for a BATCH of utterances laid back to back: utterance u owns frames utt_off[u]..utt_off[u+1]) (utt_off is
 * a HOST array of nutt+1 ints, utt_off[0] = 0).  The boundaries matter to one configuration only: gprune heu / beam
 * over tied-mixture codebooks, whose thresholds at frame t come from the codebook's winners at frame t-1 of the SAME
 * input (calc_tied_mix.c:203-215; outprob_prepare() clears the cache per input, outprob.c:138-166);
 * jamd_gmm_outprob_dev() treats its T frames as one utterance. */
int  jamd_gmm_outprob_utts_dev(jamd_gmm *g, const float *dev_frames, const int *utt_off, int nutt, float *dev_out,
                               void *stream);
/* Same with host buffers (packs HTK_Param rows, uploads, scores, downloads). */
int  jamd_gmm_outprob_host(jamd_gmm *g, const float *host_frames, int T, float *host_out);
/* Tied-mixture codebook cache of calc_tied_mix.c:189-227 for inspection: top-N
 * (score,id) per (frame, codebook), descending, as MIXCACHE (hmm_calc.h:57-60).
 * dev_score/dev_id are [T][nbook][jamd_gmm_tmix_cap()]; dev_num [T][nbook]. */
int  jamd_gmm_tmix_cache_dev(jamd_gmm *g, const float *dev_frames, int T,
                             float *dev_score, int *dev_id, int *dev_num, void *stream);
/* Slots per (frame, codebook) of that cache (gprune_num for safe, the largest
 * codebook for none; 0 when the model has no tied-mixture state) and the
 * codebook count. */
int  jamd_gmm_tmix_cap(const jamd_gmm *g);
int  jamd_gmm_nbook(const jamd_gmm *g);
/* Name of the kernel variant the last outprob call used ("tile<39,2>", ...). */
const char *jamd_gmm_last_kernel(const jamd_gmm *g);
/* Per-Gaussian scores for Julius' official plugin slot (compute_gaussset: calcmix(),
 * plugin/calcmix.c:86-323): out[t][e] = (gconst_e + sum_d (o_td - mean_ed)^2 * ivar_ed) * -0.5
 * for every mixture entry e (state order, e = st_off[s] + i for component i of state s), LOG_ZERO
 * for a NULL density -- what the sample plugin's calcmix() stores in OP_calced_score[i]; calc_mix()
 * then adds the weights and takes the log-sum itself.  out is [T][jamd_gmm_nentry()] row-major.
 * Single-stream models whose states are all plain or all tied-mixture. */
int  jamd_gmm_nentry(const jamd_gmm *g);
/* For a model whose states are all tied-mixture the columns are the codebook Gaussians instead
 * (what calc_tied_mix() sends through the slot: book->d, libsent/src/phmm/calc_tied_mix.c:189-227):
 * column jamd_gmm_book_offsets()[b] + k is Gaussian k of codebook b; off receives nbook + 1 ints. */
int  jamd_gmm_book_offsets(const jamd_gmm *g, int *off, int cap);
int  jamd_gmm_dens_dev(jamd_gmm *g, const float *dev_frames, int T, float *dev_out, void *stream);
int  jamd_gmm_dens_host(jamd_gmm *g, const float *host_frames, int T, float *host_out);

/* ---- Gaussian mixture selection (-gshmm FILE -gsnum N) --------------------------------------
 * Replaces gms_state() (libsent/src/phmm/gms.c:394-412, with compute_gs_scores()/do_gms(),
 * gms.c:189-264, and compute_g_max(), gms_gprune.c:80-150).  gs is the flattened selection model
 * (plain single-stream GMM states, one per GS HMM state in gms.c:106-117's order), state2gs[s] the
 * selection state of state s of the real model (state2gs in gms.c:119-145; -1 = state not mapped,
 * left alone), nbest the -gsnum value.  jamd_gms_apply_dev() turns a [T][nstate] matrix of real
 * scores into what outprob_state() returns under GMS: where the selection state of s is not among
 * the frame's nbest, scores[t][s] becomes the selection state's own score.  utt_off[nutt+1] (host)
 * gives the first frame of every utterance in the T concatenated frames (the reference resets the
 * last-best Gaussian at every utterance, gms_gprune.c:190-207); NULL = one utterance. */
typedef struct jamd_gms jamd_gms;
int  jamd_gms_create(jamd_engine *e, const jamd_gmm_desc *gs, const int *state2gs, int nstate, int nbest,
                     jamd_gms **out);
/* A selection model file written by jamd_export (-gshmm given): jamd_gms_save(),
 * julius_amd/shim/jamd_flatten.c. */
int  jamd_gms_load(jamd_engine *e, const char *path, jamd_gms **out);
/* on != 0: select with the reference's heap (sort_gsindex_upward(), gms.c:189-230) on one lane, so
 * that an exact tie between two selection states on the nbest boundary falls as in the reference;
 * default is the wave-parallel ranking, where the lower state id wins such a tie. */
int  jamd_gms_set_strict_order(jamd_gms *m, int on);
int  jamd_gms_nstate(const jamd_gms *m);     /* states of the REAL model (columns of the score matrix) */
void jamd_gms_destroy(jamd_gms *m);
int  jamd_gms_apply_dev(jamd_gms *m, const float *dev_frames, int T, const int *utt_off, int nutt,
                        float *dev_scores, void *stream);
/* The same over host buffers (frames in, scores in and out), on the engine's own stream. */
int  jamd_gms_apply_host(jamd_gms *m, const float *host_frames, int T, const int *utt_off, int nutt,
                         float *host_scores);

/* ---- GMM-based input verification / rejection (-gmm FILE -gmmnum N -gmmreject NAMES) ---------
 * Replaces the scoring inside gmm_proceed() (libjulius/src/gmm.c:574-600; gmm_calc_mix() :335-370,
 * gmm_gprune_safe() :296-313 with gmm_compute_g_base() :177-194 / gmm_compute_g_safe() :218-240):
 * per frame, the log10 likelihood of each verification GMM's single output state under gmm.c's own
 * top-N pruning -- which is not libsent's arithmetic (gconst is added last for the first N
 * Gaussians), so this is a separate kernel, not jamd_gmm_outprob_*().  gmm is the flattened GMM
 * definition file (jamd_flatten_hmminfo(recog->gmm)), model_state[k] the state id of model k's
 * output state (d->s[1]->id in recog->gmm->start order, the order of gc->gmm_score[]),
 * gprune_num the -gmmnum value (jconf->reject.gmm_gprune_num, default 10).
 *   frame scores  [T][nmodel]     what one gmm_proceed() adds to gc->gmm_score[k]
 *   utt scores    [nutt][nmodel]  gc->gmm_score[] at gmm_end(): float sums in frame order
 * gmm_end()'s winner / confidence / gmm_valid_input() stay with the caller (a few flops). */
typedef struct jamd_rejgmm jamd_rejgmm;
int  jamd_rejgmm_create(jamd_engine *e, const jamd_gmm_desc *gmm, const int *model_state, int nmodel,
                        int gprune_num, jamd_rejgmm **out);
void jamd_rejgmm_destroy(jamd_rejgmm *m);
/* A file written by jamd_export (-gmm given: jamd_rejgmm_save(), julius_amd/shim/jamd_flatten.c):
 * the GMMs, -gmmnum, the model names and which of them -gmmreject names. */
int  jamd_rejgmm_load(jamd_engine *e, const char *path, jamd_rejgmm **out);
/* Names and gc->is_voice[] (gmm_init(), gmm.c:466-480: 0 for models named by -gmmreject) for a handle
 * made with jamd_rejgmm_create(); jamd_rejgmm_load() sets them from the file. */
int  jamd_rejgmm_set_models(jamd_rejgmm *m, const char *const *names, const unsigned char *is_voice);
const char *jamd_rejgmm_model_name(const jamd_rejgmm *m, int k);
/* gmm_end() (gmm.c:614-660) + gmm_valid_input() (:672-679) on one row of utterance sums (host
 * arithmetic, a few flops): the winning model, its confidence 1 / sum_i 10^(0.05 (score_i - max)), and
 * whether the input is accepted (the winner is not a rejected model). */
int  jamd_rejgmm_verdict(const jamd_rejgmm *m, const float *utt_scores, int *winner, float *cm, int *accepted);
int  jamd_rejgmm_nmodel(const jamd_rejgmm *m);
int  jamd_rejgmm_veclen(const jamd_rejgmm *m);
int  jamd_rejgmm_frame_scores_dev(jamd_rejgmm *m, const float *dev_frames, int T, float *dev_out, void *stream);
int  jamd_rejgmm_utt_scores_dev(jamd_rejgmm *m, const float *dev_frame_scores, int T, const int *utt_off, int nutt,
                                float *dev_out, void *stream);
/* Host buffers in and out; either output may be NULL (utt_off/nutt only needed for utt scores). */
int  jamd_rejgmm_scores_host(jamd_rejgmm *m, const float *host_frames, int T, const int *utt_off, int nutt,
                             float *host_frame_scores, float *host_utt_scores);

/* ------------------------------------------------- pseudo-phone state sets */
/* CD_State_Set table (htk_hmm.h:249-253): set i = states[set_off[i]..set_off[i+1]).
 * Replaces outprob_cd() (outprob.c:383): cd[t][i] from one [T][nstate] score
 * matrix. nbest = hmminfo->cdmax_num. */
int  jamd_cdset_create(jamd_engine *e, int nset, const int *set_off, const int *states,
                       int method, int nbest, jamd_cdset **out);
void jamd_cdset_destroy(jamd_cdset *c);
int  jamd_cdset_nset(const jamd_cdset *c);
int  jamd_cdset_outprob_dev(jamd_cdset *c, const float *dev_scores, int T, int nstate,
                            float *dev_cd, void *stream);

/* --------------------------------------------------------------------- DNN */
/* Flattened DNNData (libsent/include/sent/dnn.h:41-74): nlayer = hnum + 1
 * affine layers, dims[0..nlayer] = inputnodenum, hidden..., outputnodenum;
 * w[l] is [dims[l+1]][dims[l]] row-major exactly as dnn_layer_load() reads the
 * .npy (calc_dnn.c:225-335), b[l] is [dims[l+1]]; state_prior[dims[nlayer]] as
 * stored after dnn_setup() (log10 already applied when state_prior_log10nize,
 * calc_dnn.c:699-703). */
typedef struct {
  int nlayer;
  const int *dims;
  const float *const *w;
  const float *const *b;
  const float *state_prior;
} jamd_dnn_desc;

int  jamd_dnn_create(jamd_engine *e, const jamd_dnn_desc *d, jamd_dnn **out);
/* The same from Julius' own on-disk DNN definition: the -dnnconf text file, the NumPy
 * .npy weight / bias files and the state prior list it names, read as the reference does
 * (dnn_config_file_parse(), libjulius/src/m_jconf.c:577-735; load_npy() / dnn_layer_load()
 * / prior file, libsent/src/phmm/calc_dnn.c:225-335, :390-434, :678-707).  Relative
 * paths are relative to the dnnconf file. */
int  jamd_dnn_load(jamd_engine *e, const char *dnnconf_path, jamd_dnn **out);
void jamd_dnn_destroy(jamd_dnn *n);
int  jamd_dnn_nstate(const jamd_dnn *n);   /* dims[nlayer] */
int  jamd_dnn_veclen(const jamd_dnn *n);   /* dims[0]      */
/* Replaces dnn_calc_outprob() (calc_dnn.c:774) for a batch of frames:
 * out[t][i] = INV_LOG_TEN*(x_i - addlog_array(x)) - state_prior[i]
 * (calc_dnn.c:858-866).  frames [T][dims[0]] already spliced
 * (libjulius/src/wav2mfcc.c:163-169 does the splicing upstream). */
int  jamd_dnn_outprob_dev(jamd_dnn *n, const float *dev_frames, int T, float *dev_out,
                          void *stream);
int  jamd_dnn_outprob_host(jamd_dnn *n, const float *host_frames, int T, float *host_out);


/* ------------------------------------------------- first-pass beam (pass 1) */
/* Flattened read side of the first pass: the tree lexicon WCHMM_INFO
 * (libjulius/include/julius/wchmm.h:211-278), the cross-word context tables
 * that outprob_style() (libjulius/src/outprob_style.c:354) resolves by name
 * lookups, the LM factoring tables (libjulius/src/factoring_sub.c:345-468) and
 * the forward 2-gram the beam reads through ngram->bigram_prob
 * (libsent/src/ngram/ngram_access.c:288-403).  N-gram LM with 1-gram factoring (the
 * reference's default "fast" setup), DFA grammar with per-category trees, or isolated word list; multipath models
 * through jamd_flatten_lexicon_multipath() (JAMD_LM_MULTIPATH below).  All indices are
 * 32-bit; WORD_INVALID is -1 here.  Built by jamd_flatten_lexicon()
 * (julius_amd/shim/jamd_flatten_lex.c) from an unmodified RecogProcess.
 * NOT SERVED -- the flattener returns JAMD_EINVAL, the shim's get_back_trellis_init() logs the reason and returns FALSE
 * (there is no CPU first pass behind this library):
 *   - a grammar without per-category trees (not a runtime choice of the reference: multigram_build() sets
 *     wchmm->category_tree = TRUE for every grammar, multi-gram.c:101), and user-defined LM functions (LM_NGRAM_USER);
 *   - N-gram lexicons built without 1-gram factoring (a non-default ./configure of the reference). */
#define JAMD_AS_STATE 0   /* AS_STATE  wchmm.h:105: out_id = state id                   */
#define JAMD_AS_LSET  1   /* AS_LSET   wchmm.h:106: out_id = state-set id                */
#define JAMD_AS_RSET  2   /* AS_RSET   wchmm.h:107: out_id = row of lc_tab               */
#define JAMD_AS_LRSET 3   /* AS_LRSET  wchmm.h:108: out_id = row of lc_tab               */
#define JAMD_AS_NONE  4   /* multipath only: non-emitting node (state[n].out.state == NULL) */

#define JAMD_LM_NGRAM 0   /* LM_PROB  */
#define JAMD_LM_DFA   1   /* LM_DFA, LM_DFA_GRAMMAR with the default per-category tree */
#define JAMD_LM_WORD  2   /* LM_DFA, LM_DFA_WORD: isolated word recognition (-w): every word of the list starts
                           * with a token (beam.c:1762-1788), no cross-word transition (:2810, :2875), the
                           * result is the best word on the last frame (find_1pass_result_word(), :561).
                           * Uses ninit / init_node / init_lscore (all 0.0); cat_pair is not read. */

/* OR-ed into lm_type by jamd_flatten_lexicon_multipath(): a multipath lexicon (hmminfo->multipath:
 * non-emitting word-begin / word-end nodes, word_head[] = wchmm->wordbegin[], wordend_a unused,
 * the frame loop of beam.c:2747-2836: word-internal transitions, the beam over the NEW tokens, cross-word transitions
 * from the word ends among them with the root expanded along its own arcs inside the frame, output probabilities on
 * emitting nodes, the final cut).  Decoded by the exact-order kernel's multipath frame (csrc/beam_exact_mp.h: one
 * workgroup per utterance, either workgroup shape, streaming included) in the reference's own tie order, or by the strict-order
 * kernel (JAMD_ORDER_STRICT); JAMD_ORDER_FAST does not take them.  One kind of lexicon is strict-order only: a root
 * that reaches a word-end node along its own arcs (a word made of tee models only) -- jamd_beam_order_mode() then
 * reports JAMD_ORDER_FAST for the new work area and jamd_beam_set_order_mode(b, JAMD_ORDER_EXACT) says why.  The reference
 * never builds such a lexicon (wchmm_add_word() rejects the word: "WORD SKIPPING TRANSITION NOT ALLOWED", wchmm.c:1345-1362;
 * tests/test_beam_oracle.py shows it for a grammar and for an N-gram), so jamd_export / the shim cannot produce one: the
 * restriction concerns hand-written descriptors only. */
#define JAMD_LM_MULTIPATH 0x100

#define JAMD_NG_NORMAL         0  /* bi_prob_normal()            ngram_access.c:288 */
#define JAMD_NG_ADDITIONAL_OLD 1  /* bi_prob_additional_oldbin() ngram_access.c:320 */
#define JAMD_NG_ADDITIONAL     2  /* bi_prob_additional()        ngram_access.c:351 */
#define JAMD_NG_COMPUTE        3  /* bi_prob_compute()           ngram_access.c:383 */

typedef struct {
  /* tree lexicon */
  int nnode, nword, startnum, isolatenum;
  const float *self_a, *next_a;        /* [nnode] wchmm->self_a / next_a (next goes to node+1)   */
  const int   *ac_off;                 /* [nnode+1] extra arcs (A_CELL2 chains, wchmm.h:162) in   */
  const int   *ac_to;                  /*           the order beam_intra_word() walks them        */
  const float *ac_a;                   /*           (libjulius/src/beam.c:2173-2177)              */
  const int   *stend;                  /* [nnode] word ending here or -1 (wchmm->stend)           */
  const int   *scid;                   /* [nnode] wchmm->state[n].scid                            */
  const unsigned char *out_kind;       /* [nnode] JAMD_AS_*  (wchmm->outstyle)                    */
  const int   *out_id;                 /* [nnode] see JAMD_AS_*                                   */
  /* cross-word left context: column = base phone of the previous word's last
   * phone (center_name(), libsent/src/hmminfo/cdhmm.c:144); column nlc = no
   * previous word.  Entry >= 0: state id; < 0: ~(state-set id). */
  int nlc, nlcrow;
  const int   *lc_tab;                 /* [nlcrow][nlc+1]                                         */
  const int   *word_lc;                /* [nword] column a word selects as LEFT context           */
  /* pseudo-phone state sets (CD_State_Set) used by AS_LSET nodes and lc_tab */
  int nset;
  const int   *set_off, *set_states;   /* CSR                                                     */
  int cdset_method, cdmax_num;         /* JAMD_IWCD_*, hmminfo->cdmax_num                         */
  /* tree roots */
  const int   *startnode;              /* [startnum] wchmm->startnode                             */
  const int   *start2isolate;          /* [startnum] wchmm->start2isolate (-1 = shared root)      */
  /* words */
  const float *wordend_a;              /* [nword] wchmm->wordend_a                                */
  const int   *wton;                   /* [nword] winfo->wton (N-gram entry id)                   */
  const float *cprob;                  /* [nword] winfo->cprob                                    */
  const unsigned char *is_transparent; /* [nword]                                                 */
  const int   *word_head;              /* [nword] wchmm->offset[w][0]                             */
  int head_silwid, tail_silwid;
  /* LM factoring */
  int nfscore, nscword;
  const float *fscore;                 /* [nfscore] wchmm->fscore (index -scid)                   */
  const int   *scword;                 /* [nscword] wchmm->scword (index scid)                    */
  /* forward 2-gram as ngram->bigram_prob reads it */
  int ng_mode;                         /* JAMD_NG_*                                               */
  int ng_nword, ng_nbigram, ng_unk_id;
  float ng_unk_num_log;
  const float *ng_uni_prob;            /* d[0].prob                                               */
  const float *ng_uni_bo;              /* d[0].bo_wt, or bo_wt_1 for the ADDITIONAL modes         */
  const int   *ng_bi_bgn;              /* d[1].bgn  (-1 = NNID_INVALID)                           */
  const int   *ng_bi_num;              /* d[1].num                                                */
  const int   *ng_bi_wid;              /* d[1].nnid2wid                                           */
  const float *ng_bi_prob;             /* d[1].prob, or p_2 for the ADDITIONAL modes              */
  /* FSBeam local copies (libjulius/include/julius/recog.h:147-150) */
  float lm_weight, lm_penalty, lm_penalty_trans;
  /* ---- grammar (DFA) mode, lm_type == JAMD_LM_DFA: per-category tree lexicon
   * (wchmm->category_tree), category-pair constraint at the word boundaries
   * (beam_inter_word(), libjulius/src/beam.c:2404-2412; dfa_cp(), libsent/src/dfa/cpair.c),
   * no LM factoring inside words.  For N-gram mode these are 0 / NULL.  In grammar mode
   * `wton` holds the category of each word. */
  int lm_type;                         /* JAMD_LM_NGRAM / JAMD_LM_DFA / JAMD_LM_WORD               */
  int ncat;                            /* dfa->term_num                                            */
  const unsigned char *cat_pair;       /* [ncat][ncat] dfa_cp(dfa, c1, c2): c2 may follow c1       */
  const int   *start2wid;              /* [startnum] wchmm->start2wid (a word of the root's tree)  */
  int ninit;                           /* initial tokens of init_nodescore() (beam.c:1669-1757):   */
  const int   *init_node;              /* [ninit] distinct word-head nodes in creation order       */
  const float *init_lscore;            /* [ninit] penalty1 + cprob of the first word reaching it   */
  float penalty1;                      /* r->config->lmp.penalty1                                  */
  /* ---- a FORWARD DFA beside the reversed one (the `.dfa.forward` file recent mkdfa.pl writes next to `.dfa`,
   * libjulius/src/multi-gram.c:868-880).  Tokens then carry a state of it (TOKEN2.to_state): an initial token of
   * category t takes the arc labelled t out of the grammar's first state (libjulius/src/beam.c:1739-1747), a cross-word
   * transition to a root of category c takes the arc labelled c out of the token's state and is dropped when there is
   * none (:2412-2422), word-internal transitions inherit it (:2120).  nfwd = 0: no forward DFA (all pointers may be NULL).
   * Served by the exact-order kernels (JAMD_ORDER_EXACT, the default; multipath lexicons included) and the strict-order
   * kernels; the canonical-tie kernel (JAMD_ORDER_FAST) carries no such state and is refused for these lexicons. */
  int nfwd;                            /* states of wchmm->dfa_forward                              */
  const int   *fwd_off;                /* [nfwd+1] arcs of a state, in the order of its arc list     */
  const int   *fwd_label;              /*          arc label = word category                         */
  const int   *fwd_to;                 /*          next state                                        */
  const int   *init_to_state;          /* [ninit] to_state of initial token e (-1: no such arc)      */
} jamd_lexicon_desc;

/* One emitted word-trellis record (TRELLIS_ATOM, libjulius/include/julius/
 * trellis.h:28-41) as save_trellis() fills it (beam.c:2209-2247).  last_tre is
 * the index of the predecessor atom in the same output array, -1 for the
 * sentence-start sentinel (FSBeam.bos). */
typedef struct {
  int   wid;
  int   last_tre;
  float backscore;
  float lscore;
  short begintime;
  short endtime;
} jamd_trellis_atom;


typedef struct jamd_lexicon jamd_lexicon;
typedef struct jamd_beam    jamd_beam;

/* Upload the first-pass tables.  Replaces, for the device, what
 * get_back_trellis_init() reads from r->wchmm (libjulius/src/beam.c:1825).
 * JAMD_EINVAL for inconsistent tables (a root without factoring value / successor word). */
int  jamd_lexicon_create(jamd_engine *e, const jamd_lexicon_desc *d, jamd_lexicon **out);
/* Julius' BINARY model files read directly, without a Julius process (julius_amd/csrc/readers.hip):
 *   jamd_gmm_load_binhmm()   the binary HMM definition of mkbinhmm (libsent/src/hmminfo/read_binhmm.c:756): the
 *                            same device model as jamd_export's PREFIX.am made from that file;
 *   jamd_binhmm_to_blob()    the same conversion to a "JAMDGMM1" file (byte for byte jamd_export's; no device);
 * The tree lexicon is the output of libjulius/src/wchmm.c over dictionary + HMMList + LM (its factoring values and
 * the words kept out of the tree are functions of the LM), not a file format: PREFIX.lex -- tree and N-gram tables
 * together -- comes from jamd_export, which links Julius' own loaders. */
int  jamd_gmm_load_binhmm(jamd_engine *e, const char *binhmm_path, int gprune, int gprune_num, jamd_gmm **out);
int  jamd_binhmm_to_blob(const char *binhmm_path, const char *blob_path);

/* The same from a "JAMDLEX1" file written by jamd_lexicon_save()
 * (julius_amd/shim/jamd_flatten_lex.c) in a process that loaded the dictionary and LM
 * with Julius' own readers. */
int  jamd_lexicon_load(jamd_engine *e, const char *path, jamd_lexicon **out);
/* The same with the N-gram half read DIRECTLY from a binary N-gram file (mkbingram v5 -- what `-d` names;
 * libsent/src/ngram/ngram_read_bin.c:240-365,603): the 1-gram / 2-gram tables of the first pass and the choice of the
 * 2-gram (forward, or the additional forward 2-gram of a backward N-gram, ngram_access.c:449-466) come from the file,
 * the cross-word LM table is built from them; the tree half -- nodes, factoring values, word -> N-gram ids, class
 * probabilities -- stays PREFIX.lex's, which is the output of libjulius/src/wchmm.c over the dictionary and not a file
 * format.  The two must share their vocabulary (names in N-gram id order: PREFIX.lex records it; refused otherwise).
 * With the N-gram the tree was built from the result equals jamd_lexicon_load()'s.  A RETRAINED N-gram over the same
 * vocabulary: the tree depends on the 1-gram in two places -- which words wchmm.c keeps out of the tree (the -sepnum most
 * frequent ones, libjulius/src/wchmm.c:1470, :1882) and the 1-gram factoring value of every shared node
 * (libjulius/src/factoring_sub.c:429-463).  The loader recomputes the factoring values from the file's 1-gram (the best
 * uni_prob + class probability over the words below the node) when the same words stay out of the tree -- the result is
 * then the lexicon a fresh jamd_export with that N-gram writes -- and refuses (JAMD_EINVAL, "export the lexicon again")
 * when the tree itself would differ or PREFIX.lex does not record -sepnum (written before this check existed). */
int  jamd_lexicon_load_ngram(jamd_engine *e, const char *path, const char *bingram_path, jamd_lexicon **out);
/* Host only: do PREFIX.lex and a binary N-gram share their vocabulary (JAMD_OK, else JAMD_EINVAL with the reason), and
 * -- *same_tables, may be NULL -- are the file's first-pass tables byte for byte the ones PREFIX.lex holds? */
int  jamd_bingram_check(const char *lex_path, const char *bingram_path, int *same_tables);
/* Host only: the 1-gram factoring values jamd_lexicon_load_ngram() would use -- fscore[0..*nfscore) (at most `cap` are
 * written; index 0 is unused, as in WCHMM_INFO.fscore) -- with the same acceptance rules and errors. */
int  jamd_bingram_fscore(const char *lex_path, const char *bingram_path, float *fscore, int cap, int *nfscore);
void jamd_lexicon_destroy(jamd_lexicon *l);

/* First-pass status of one utterance */
#define JAMD_PASS1_OK        0  /* J_RESULT_STATUS_SUCCESS                                   */
#define JAMD_PASS1_FAIL      1  /* no sentence-end word survived (find_1pass_result(),        */
                                /* beam.c:424-429 -> J_RESULT_STATUS_FAIL)                     */
#define JAMD_PASS1_DIED      2  /* "no nodes left in beam" at frame died_at: _proceed()       */
                                /* returned FALSE (beam.c:3012-3015); the caller segments     */
#define JAMD_PASS1_OVERFLOW  3  /* more trellis atoms than atoms_per_utt                      */

typedef struct {
  int   status;          /* JAMD_PASS1_*                                              */
  int   natom;           /* trellis atoms emitted                                     */
  int   wnum;            /* r->pass1_wnum                                             */
  float score;           /* r->pass1_score (best->backscore, beam.c:505)              */
  int   died_at;         /* frame index for JAMD_PASS1_DIED, else -1                  */
  int   ties;            /* exact score ties met by a max/selection:  */
                         /* 0 => the result does not depend on visiting order         */
  int   frames;          /* T                                                         */
  int   max_tokens;      /* high-water mark of tokens alive in one frame              */
  int   ties_node, ties_wordend, ties_cut;   /* ties by kind: Viterbi max at a node, best    */
                         /* word end, rank cut                                        */
  int   phase_us[8];     /* device time spent in: A intra-word+atoms, B cross-word,   */
                         /* C finalize+outprob, D rank pruning (microseconds); [4..6] */
                         /* thread 0's share of C: key fetch, payload, outprob        */
  int   wseq[150];       /* r->pass1_wseq, MAXSEQNUM = 150 (libsent speech.h:50)      */
} jamd_pass1_result;

/* Work area for up to max_utts utterances decoded at once (FSBeam,
 * libjulius/include/julius/recog.h:115-174, one per utterance).
 * beam_width = r->trellis_beam_width (<= 65536), score_pruning_width =
 * r->config->pass1.score_pruning_width (< 0 disables, beam.c:2954-2960),
 * atoms_per_utt bounds the word trellis of one utterance. */
int  jamd_beam_create(jamd_engine *e, jamd_lexicon *l, int beam_width, float score_pruning_width,
                      int max_utts, int atoms_per_utt, jamd_beam **out);
void jamd_beam_destroy(jamd_beam *b);

/* The whole first pass for nutt utterances at once: for each utterance
 * get_back_trellis_init() (beam.c:1825), get_back_trellis_proceed() for
 * t = 1..T-1 (beam.c:2663), get_back_trellis_end() (beam.c:3052) and
 * find_1pass_result() (beam.c:372).  dev_scores holds the state score rows of
 * all utterances back to back: utterance u owns rows utt_off[u]..utt_off[u+1])
 * of [.][nstate] (what outprob_state() would return for every state: the output
 * of jamd_gmm_outprob_dev / jamd_dnn_outprob_dev).  utt_off is a HOST array of
 * nutt+1 ints.  Asynchronous on `stream`; results are read with
 * jamd_beam_results() / jamd_beam_trellis(), which synchronise.
 * ONE launch per work area at a time: the utterance table, the slices and the result records of a jamd_beam belong to
 * its latest launch, so the next jamd_beam_pass1_dev() / jamd_beam_stream_push_dev() on the same work area goes on the
 * SAME stream (it then queues behind the first) or after the first has completed; work areas are independent of each
 * other.  The same holds for the scoring calls of one jamd_gmm with history pruning (utterance boundaries are staged per
 * model). */
int  jamd_beam_pass1_dev(jamd_beam *b, const float *dev_scores, int nstate, const int *utt_off,
                         int nutt, void *stream);
/* Streaming form of jamd_beam_pass1_dev() for input that arrives in pieces (Julius calls
 * get_back_trellis_proceed() once per frame, libjulius/src/pass1.c:242; live audio never has
 * the whole utterance).  jamd_beam_stream_begin() opens a session for utterance slots
 * 0..nutt-1; each jamd_beam_stream_push_dev() advances every utterance by the rows
 * chunk_off[u]..chunk_off[u+1]) of dev_scores (any number of frames, zero included) and keeps
 * the search state on the device; `final` != 0 additionally runs get_back_trellis_end() and the
 * traceback, after which jamd_beam_results()/jamd_beam_trellis() return exactly what one
 * jamd_beam_pass1_dev() call over the concatenated rows would.  jamd_beam_results() may also
 * be called between pushes (natom / frames so far).  In strict-order mode a session must
 * consist of one final push. */
int  jamd_beam_stream_begin(jamd_beam *b, int nutt);
int  jamd_beam_stream_push_dev(jamd_beam *b, const float *dev_scores, int nstate, const int *chunk_off,
                               int nutt, int final, void *stream);

/* Verification mode.  on != 0: later jamd_beam_pass1_dev() calls run the reference's
 * SEQUENTIAL algorithm (same token creation order, same partial heap sort
 * beam.c:1342-1516, first-writer-wins propagation) with one lane per utterance, so that
 * the word trellis equals the reference's bit for bit even where exact score ties are
 * broken by visiting order.  Two to three orders of magnitude slower per utterance;
 * parallel only across the utterances of a batch.  Default is off (the frame-parallel
 * kernel, identical whenever jamd_pass1_result.ties == 0). */
int  jamd_beam_set_strict_order(jamd_beam *b, int on);
/* How exact score ties are resolved (they are the only freedom a parallel schedule has; every
 * float is the reference's float in all modes):
 *   JAMD_ORDER_EXACT   (default; multipath lexicons included; beams up to about 12 000: up to ~950 the survivors live
 *       in LDS; wider beams keep them in the utterance's slice of HBM and the pruning step overlays the whole LDS
 *       image; the closed-form extraction serves beams up to 4 400, beyond that the heap's extraction loop itself
 *       runs pipelined on one wave):
 *       frame-parallel kernel with the reference's own semantics -- candidates keyed by their position
 *       in the reference's visiting order (first writer wins, propagate_token() beam.c:1945-1980,
 *       wordend_best :2308), token creation order, and the partial heap sort of
 *       sort_token_no_order() (beam.c:1342-1516) reproduced exactly.  The word trellis equals the
 *       reference's bit for bit, ties included.
 *   JAMD_ORDER_EXACT_SERIAL  the same kernel with the heap's extraction loop run sequentially on one
 *       lane instead of in closed form / pipelined (cross-check and timing).
 *   JAMD_ORDER_FAST    frame-parallel kernel with canonical tie breaks (larger source id, smaller node
 *       on the rank cut): identical to the reference whenever jamd_pass1_result.ties == 0.
 *   JAMD_ORDER_STRICT  = jamd_beam_set_strict_order(b, 1): the sequential algorithm, one lane per utterance.
 * jamd_beam_set_strict_order(b, 0) returns to the default of the work area. */
#define JAMD_ORDER_FAST 0
#define JAMD_ORDER_STRICT 1
#define JAMD_ORDER_EXACT 2
#define JAMD_ORDER_EXACT_SERIAL 3
int  jamd_beam_set_order_mode(jamd_beam *b, int mode);
int  jamd_beam_order_mode(const jamd_beam *b);
/* Workgroup shape of the exact-order kernel (a scheduling choice: the results are the same bit for bit).
 *   JAMD_SHAPE_FULL  one utterance per CU: 1024 threads and the CU's whole LDS -- the lowest latency per utterance.
 *   JAMD_SHAPE_HALF  512 threads and half the LDS, so two utterances share a CU and the barriers and wave-serial
 *       sections of one overlap the work of the other: more frames per second once a launch carries more
 *       utterances than the device can hold in the full shape.  Available while half a CU's LDS still holds a
 *       typical frame (beams up to about 1 000, a little more with few word-initial nodes); JAMD_ESTATE otherwise.
 *   JAMD_SHAPE_AUTO  (default) HALF for launches (and streaming sessions) of more than 1.5 x the CU count
 *       utterances when available, else FULL.
 * jamd_beam_workgroup_shape() tells which one a launch of nutt utterances would use in the exact order modes (the
 * canonical-tie and strict-order kernels have one shape each and ignore the setting).  A streaming session keeps the
 * shape it was opened with. */
#define JAMD_SHAPE_AUTO 0
#define JAMD_SHAPE_FULL 1
#define JAMD_SHAPE_HALF 2
int  jamd_beam_set_workgroup_shape(jamd_beam *b, int shape);
int  jamd_beam_workgroup_shape(const jamd_beam *b, int nutt);
/* Which LDS image the exact-order kernel uses for this work area in its full shape: 0 = it cannot serve the work area,
 * 1 = narrow (survivors in LDS: beams up to ~950), 2 = wide (survivors in the utterance's slice; the sweep replay and the
 * closed form of the downward sort live in this one).  Same results either way; for tests and diagnostics. */
int  jamd_beam_exact_layout(const jamd_beam *b);
/* Blocks the host until everything queued ahead of the latest first-pass launch of this work area has completed, i.e.
 * until that kernel is next to run (returns at once when nothing was launched).  For hosts that pipeline batches: a
 * first pass that fills the device (one or two workgroups per CU, all of a CU's LDS) must get its workgroups placed
 * before the scoring kernels of the NEXT batch are queued on another stream -- queued earlier they take the LDS the
 * first pass wants, and the first pass of 512 utterances takes twice as long; queued once it runs they only fill the
 * CUs its shorter utterances leave (julius_amd/host/jamd_batch.c, bench.py: -5 % per step). */
int  jamd_beam_wait_started(jamd_beam *b);
/* The same ordering WITHOUT the host: work queued on `stream` after this call starts only when the workgroups of the
 * latest first-pass launch of this work area that fit the device at once (one per CU; two in the exact-order kernel's
 * half shape) have started -- they bump a counter in signal memory as their first instruction and `stream`'s command
 * processor waits on it (hipStreamWaitValue32).  A pipelining host calls it between the first-pass launch of batch k
 * and the scoring launches of batch k+1 (jamd_batch.c, bench.py): no event wait, no sleep, nothing to time.  Devices
 * without wait-on-memory fall back to jamd_beam_wait_started() plus a millisecond's pause inside this call. */
int  jamd_beam_stream_wait_resident(jamd_beam *b, void *stream);
/* Test entry: sets the resident counter (the device word the first-pass workgroups bump, and the host's bookkeeping of
 * it) to `count`, as if that many workgroups had been launched since the work area was created.  The counter is 32 bits
 * wide and drained back to zero before it could wrap (csrc/beam.hip mark_started()); a test starts it just below that
 * point instead of launching 2^31 workgroups.  The device is synchronised first. */
int  jamd_beam_debug_preset_resident(jamd_beam *b, unsigned count);
/* Test entry: workgroups accounted since the last drain (*launched) and the value the next
 * jamd_beam_stream_wait_resident() waits for (*target). */
int  jamd_beam_debug_resident(const jamd_beam *b, unsigned *launched, unsigned *target);
/* The rank-pruning step alone (sort_token_no_order(), beam.c:1492): given the scores of the n tokens of
 * a frame in creation order (host array), writes the token indices the next frame visits, in visiting
 * order (tindex[n_start..n_end]), for the work area's beam width; *nkeep = how many.  Runs the
 * exact-order kernel's pruning code on the device; diagnostic / test entry. */
int  jamd_beam_prune_order(jamd_beam *b, const float *scores, int n, int *order, int *nkeep);
/* The same step with the WHOLE array out: tindex[0..n) = the token ids at array positions 0..n-1 after
 * sort_token_no_order() (residual heap and extracted part: what the multipath frame's final cut starts from,
 * csrc/beam_exact_mp.h) beside order[].  Test entry like jamd_beam_prune_order(); full workgroup shape. */
int  jamd_beam_prune_arrange(jamd_beam *b, const float *scores, int n, int *order, int *nkeep, int *tindex);
/* How the latest jamd_beam_prune_order() call resolved the events of the extraction loop (tail positions that hold a
 * top element, beam.c:1369-1383): *sweep_rounds = rounds of the all-at-once sweep replay (csrc/beam_sweep.h) when it
 * ran and converged, -1 = it ran and handed the frame to the extraction loop, 0 = not needed (few candidates, or no
 * tied element among them); *sweep_us (may be NULL) = its duration on the device, *sweep_events (may be NULL) = events
 * it held at the end.  Diagnostic / test entry. */
int  jamd_beam_prune_info(jamd_beam *b, int *sweep_rounds, int *sweep_us, int *sweep_events);
/* How the rank pruning steps (sort_token_no_order(), beam.c:1492) of utterance `utt` were resolved by the exact-order
 * kernel since the work area was created or the counters were last reset -- frames counted by path:
 *   [0] frames pruned (more tokens than the beam)          [1] upward, closed form, no tied element on a tail position
 *   [2] upward, wave-serial event replay                   [3] upward, sweep replay converged
 *   [4] upward, sweep gave the frame to the extraction loop [5] downward, closed form (sweep + residual-heap replay)
 *   [6] extraction loop itself (pipelined / serial)        [7] sweep rounds, total
 * and the work of the frames (sums over the frames): [8] tokens created, [9] survivors visited, [10] word ends among them,
 * [11] frames; [12..15] reserved.  reset != 0 clears them.  Diagnostic. */
int  jamd_beam_prune_stats(jamd_beam *b, int utt, int stats[16], int reset);
int  jamd_beam_results(jamd_beam *b, jamd_pass1_result *out, int nutt);
/* Word trellis of utterance u in emission order (last_tre indexes the same
 * array).  bt_relocate_rw()/bt_sort_rw() order (libjulius/src/backtrellis.c:
 * 218-267,468-477) is (endtime, wid). */
int  jamd_beam_trellis(jamd_beam *b, int utt, jamd_trellis_atom *atoms, int cap, int *natom);

#ifdef __cplusplus
}
#endif
#endif /* JULIUS_AMD_H */
