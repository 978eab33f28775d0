/*
 * jamd_oracle_beam.c -- CPU restatement of Julius' first pass (frame-synchronous
 * token passing over the tree lexicon).
 *
 * *** TEST INFRASTRUCTURE ONLY (see jamd_oracle.h). ***
 *
 * Written from scratch over the FLAT tables of include/julius_amd.h
 * (jamd_lexicon_desc), following the reference's sequential algorithm step by
 * step -- including the partial heap sort that decides the iteration order of
 * the surviving tokens -- so that its word trellis is the reference's word
 * trellis even where Viterbi ties are broken by visiting order.  Scope: N-gram
 * LM, DFA grammar with per-category trees or isolated-word lists, non-multipath models, the reference's default "fast" configuration
 * (UNIGRAM_FACTORING, PASS1_IWCD, SCORE_PRUNING; no WPAIR / WORD_GRAPH /
 * spsegment).  Paths below are relative to the reference root.
 *
 * Parity status: PINNED against the compiled reference (oracle/_ref) by
 * tests/test_beam_oracle.py and the fixtures under tests/golden/.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jamd_oracle.h"

typedef struct {          /* TOKEN2, libjulius/include/julius/beam.h:35-45 */
  int   last_tre;         /* index into atoms[], -1 = FSBeam.bos */
  int   last_cword;
  float last_lscore;
  float score;
  int   node;
} tok;

typedef struct {
  const jamd_lexicon_desc *lx;
  const float *sc; int S;
  tok *tlist[2]; int *tindex[2]; int tnum[2]; int maxtnum, expand_step;
  int *token;               /* node -> token id in tlist[tn], -1 = none */
  int tn, tl, n_start, n_end;
  float score_pruning_threshold, score_pruning_max;
  float wordend_best_score; int wordend_best_node, wordend_best_tre, wordend_best_last_cword;
  jamd_trellis_atom *atoms; int natom, atom_cap, overflow;
  int lmt;                  /* lx->lm_type without the multipath flag */
  int mp;                   /* multipath lexicon (hmminfo->multipath): beam.c:2747-2836, :2930-2943, :3066-3073 */
} beam;

/* ---- LM ------------------------------------------------------------------- */
/* search_bigram(), libsent/src/ngram/ngram_access.c:225-247 */
static int search_bigram(const jamd_lexicon_desc *lx, int w_context, int w)
{
  int left = lx->ng_bi_bgn[w_context], right, mid;
  if (left < 0) return -1;
  right = left + lx->ng_bi_num[w_context] - 1;
  while (left < right) {
    mid = (left + right) / 2;
    if (lx->ng_bi_wid[mid] < w) left = mid + 1; else right = mid;
  }
  return (lx->ng_bi_wid[left] == w) ? left : -1;
}

/* ngram->bigram_prob as selected by bi_prob_func_set(), ngram_access.c:288-466 */
float jo_bigram_prob(const jamd_lexicon_desc *lx, int w1, int w2)
{
  int n2; float prob;
  switch (lx->ng_mode) {
  case JAMD_NG_NORMAL: case JAMD_NG_ADDITIONAL_OLD:          /* :288 / :320 (LR index) */
    if ((n2 = search_bigram(lx, w1, w2)) >= 0) prob = lx->ng_bi_prob[n2];
    else prob = lx->ng_uni_bo[w1] + lx->ng_uni_prob[w2];
    break;
  case JAMD_NG_ADDITIONAL:                                   /* :351 (RL index) */
    if ((n2 = search_bigram(lx, w2, w1)) >= 0) prob = lx->ng_bi_prob[n2];
    else prob = lx->ng_uni_bo[w1] + lx->ng_uni_prob[w2];
    break;
  default:                                                   /* :383 bi_prob_compute */
    if ((n2 = search_bigram(lx, w2, w1)) >= 0) prob = lx->ng_bi_prob[n2];
    else prob = lx->ng_uni_bo[w2] + lx->ng_uni_prob[w1];
    prob = prob + lx->ng_uni_prob[w2] - lx->ng_uni_prob[w1];
    break;
  }
  if (w2 != lx->ng_unk_id) return prob;
  return prob - lx->ng_unk_num_log;
}

/* max_successor_prob(), libjulius/src/factoring_sub.c:942-1008 (UNIGRAM_FACTORING;
 * the per-scid cache there is a pure memo) */
static float max_successor_prob(const jamd_lexicon_desc *lx, int lastword, int node)
{
  int scid, w;
  if (lastword < 0) return 0.0f;
  scid = lx->scid[node];
  if (scid < 0) return lx->fscore[-scid];
  w = lx->scword[scid];
  return jo_bigram_prob(lx, lx->wton[lastword], lx->wton[w]) + lx->cprob[w];
}

/* one entry of max_successor_prob_iw()'s array, factoring_sub.c:1119-1143 */
static float iw_prob(const jamd_lexicon_desc *lx, int lastword, int stid)
{
  int w = lx->scword[lx->scid[lx->startnode[stid]]];
  return jo_bigram_prob(lx, lx->wton[lastword], lx->wton[w]) + lx->cprob[w];
}

/* ---- acoustic score of a node: outprob_style(), libjulius/src/outprob_style.c:354 */
static float outprob_style(const beam *b, int node, int last_wid, int t)
{
  const jamd_lexicon_desc *lx = b->lx;
  const float *row = b->sc + (size_t)t * b->S;
  int id = lx->out_id[node], ent;
  switch (lx->out_kind[node]) {
  case JAMD_AS_STATE: return row[id];
  case JAMD_AS_LSET: ent = ~id; break;
  default:
    ent = lx->lc_tab[(size_t)id * (lx->nlc + 1) + (last_wid < 0 ? lx->nlc : lx->word_lc[last_wid])];
    break;
  }
  if (ent >= 0) return row[ent];
  ent = ~ent;
  return jo_outprob_cd(row, lx->set_states + lx->set_off[ent], lx->set_off[ent + 1] - lx->set_off[ent],
                       lx->cdset_method, lx->cdmax_num);
}

/* ---- token space: beam.c:997-1200 ---------------------------------------------- */
static void expand_tlist(beam *b)                       /* :1025 */
{
  int i;
  b->maxtnum += b->expand_step;
  for (i = 0; i < 2; i++) {
    b->tlist[i] = (tok *)realloc(b->tlist[i], sizeof(tok) * b->maxtnum);
    b->tindex[i] = (int *)realloc(b->tindex[i], sizeof(int) * b->maxtnum);
  }
}
static int create_token(beam *b)                        /* :1148 */
{
  int tn = b->tn, newid = b->tnum[tn];
  b->tnum[tn]++;
  while (b->tnum[tn] >= b->maxtnum) expand_tlist(b);
  b->tindex[tn][newid] = newid;
  return newid;
}

/* sort_token_upward / _downward / _no_order, beam.c:1342-1516 (1-based heap over tindex) */
#define SD(A) ti[(A) - 1]
#define SVAL(A) (tl[ti[(A) - 1]].score)
static void sort_token_upward(beam *b, int neednum, int totalnum)
{
  tok *tl = b->tlist[b->tn]; int *ti = b->tindex[b->tn];
  int n, root, child, parent, s;
  for (root = totalnum / 2; root >= 1; root--) {
    s = SD(root); parent = root;
    while ((child = parent * 2) <= totalnum) {
      if (child < totalnum && SVAL(child) < SVAL(child + 1)) child++;
      if (tl[s].score >= SVAL(child)) break;
      SD(parent) = SD(child); parent = child;
    }
    SD(parent) = s;
  }
  n = totalnum;
  while (n > totalnum - neednum) {
    s = SD(n); SD(n) = SD(1); n--; parent = 1;
    while ((child = parent * 2) <= n) {
      if (child < n && SVAL(child) < SVAL(child + 1)) child++;
      if (tl[s].score >= SVAL(child)) break;
      SD(parent) = SD(child); parent = child;
    }
    SD(parent) = s;
  }
}
static void sort_token_downward(beam *b, int neednum, int totalnum)
{
  tok *tl = b->tlist[b->tn]; int *ti = b->tindex[b->tn];
  int n, root, child, parent, s;
  for (root = totalnum / 2; root >= 1; root--) {
    s = SD(root); parent = root;
    while ((child = parent * 2) <= totalnum) {
      if (child < totalnum && SVAL(child) > SVAL(child + 1)) child++;
      if (tl[s].score <= SVAL(child)) break;
      SD(parent) = SD(child); parent = child;
    }
    SD(parent) = s;
  }
  n = totalnum;
  while (n > totalnum - neednum) {
    s = SD(n); SD(n) = SD(1); n--; parent = 1;
    while ((child = parent * 2) <= n) {
      if (child < n && SVAL(child) > SVAL(child + 1)) child++;
      if (tl[s].score <= SVAL(child)) break;
      SD(parent) = SD(child); parent = child;
    }
    SD(parent) = s;
  }
}
/* Test tooling: with JAMD_ORACLE_DUMP_SORT=<file> every call appends (n, neednum, n scores in tindex order) -- real
 * inputs of the rank pruning step for tools/prune_lab.py and the device fuzz tests. */
static void dump_sort_input(beam *b, int neednum, int totalnum)
{
  static const char *path = NULL; static int asked = 0;
  if (!asked) { path = getenv("JAMD_ORACLE_DUMP_SORT"); asked = 1; }
  if (path && totalnum > neednum) {
    FILE *f = fopen(path, "ab");
    if (f) {
      int i, hdr[2]; hdr[0] = totalnum; hdr[1] = neednum;
      fwrite(hdr, sizeof(int), 2, f);
      for (i = 0; i < totalnum; i++) fwrite(&b->tlist[b->tn][b->tindex[b->tn][i]].score, sizeof(float), 1, f);
      fclose(f);
    }
  }
}
static void sort_token_no_order(beam *b, int neednum)   /* :1492 */
{
  int totalnum = b->tnum[b->tn], restnum = totalnum - neednum;
  dump_sort_input(b, neednum, totalnum);
  if (neednum >= totalnum) { b->n_start = 0; b->n_end = totalnum - 1; }
  else if (neednum < restnum) { sort_token_upward(b, neednum, totalnum); b->n_start = totalnum - neednum; b->n_end = totalnum - 1; }
  else { sort_token_downward(b, restnum, totalnum); b->n_start = 0; b->n_end = neednum - 1; }
}

/* sort_token_no_order() alone (beam.c:1492 over :1342-1480): n tokens with these scores in creation
 * order (tindex = identity, create_token() :1148) -> the token ids the next frame visits, in visiting
 * order tindex[n_start..n_end].  Returns how many. */
int jo_sort_token_no_order(const float *scores, int n, int beam_width, int *order)
{
  beam B; int i, k;
  memset(&B, 0, sizeof(B));
  B.tn = 0;
  B.tlist[0] = (tok *)malloc(sizeof(tok) * (n > 0 ? n : 1));
  B.tindex[0] = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
  for (i = 0; i < n; i++) { B.tlist[0][i].score = scores[i]; B.tindex[0][i] = i; }
  B.tnum[0] = n;
  sort_token_no_order(&B, beam_width);
  k = 0;
  for (i = B.n_start; i <= B.n_end; i++) order[k++] = B.tindex[0][i];
  free(B.tlist[0]); free(B.tindex[0]);
  return k;
}

/* The same with the whole array out: tindex[0..n) after the sort (what the multipath frame's second sort starts from). */
int jo_sort_token_arrange(const float *scores, int n, int beam_width, int *order, int *tindex)
{
  beam B; int i, k;
  memset(&B, 0, sizeof(B));
  B.tn = 0;
  B.tlist[0] = (tok *)malloc(sizeof(tok) * (n > 0 ? n : 1));
  B.tindex[0] = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
  for (i = 0; i < n; i++) { B.tlist[0][i].score = scores[i]; B.tindex[0][i] = i; }
  B.tnum[0] = n;
  sort_token_no_order(&B, beam_width);
  k = 0;
  for (i = B.n_start; i <= B.n_end; i++) order[k++] = B.tindex[0][i];
  for (i = 0; i < n; i++) tindex[i] = B.tindex[0][i];
  free(B.tlist[0]); free(B.tindex[0]);
  return k;
}

/* propagate_token(), beam.c:1945-1980 */
static void propagate_token(beam *b, int next_node, float next_score, int last_tre, int last_cword,
                            float last_lscore)
{
  tok *tk; int id;
  if (next_score <= JO_LOG_ZERO) return;
  if ((id = b->token[next_node]) >= 0) {
    tk = &b->tlist[b->tn][id];
    if (tk->score < next_score) {
      tk->last_tre = last_tre; tk->last_cword = last_cword; tk->last_lscore = last_lscore; tk->score = next_score;
    }
  } else {
    id = create_token(b);
    tk = &b->tlist[b->tn][id];
    tk->last_tre = last_tre; tk->last_cword = last_cword; tk->last_lscore = last_lscore; tk->score = next_score;
    b->token[next_node] = id; tk->node = next_node;      /* node_assign_token :1194 */
  }
}

/* beam_intra_word_core(), beam.c:2004-2135 (LM_PROB branch) */
static void intra_word_core(beam *b, int j, int next_node, float next_a)
{
  const jamd_lexicon_desc *lx = b->lx;
  tok tk = b->tlist[b->tl][b->tindex[b->tl][j]];       /* copy: the list may be realloc'ed */
  float tmpsum = tk.score + next_a, ngram_score_cache = JO_LOG_ZERO;
  /* with per-category trees (grammar) the whole factoring block is skipped: beam.c:2029 */
  if (b->lmt == JAMD_LM_NGRAM && next_node != tk.node && lx->scid[next_node] != 0) {
    ngram_score_cache = max_successor_prob(lx, tk.last_cword, next_node) * lx->lm_weight + lx->lm_penalty;
    tmpsum -= tk.last_lscore;
    tmpsum += ngram_score_cache;
  }
  if (ngram_score_cache == JO_LOG_ZERO) ngram_score_cache = tk.last_lscore;
  propagate_token(b, next_node, tmpsum, tk.last_tre, tk.last_cword, ngram_score_cache);
}

/* beam_intra_word(), beam.c:2154-2180 */
static void intra_word(beam *b, int j)
{
  const jamd_lexicon_desc *lx = b->lx;
  int node = b->tlist[b->tl][b->tindex[b->tl][j]].node, k;
  if (lx->self_a[node] != JO_LOG_ZERO) intra_word_core(b, j, node, lx->self_a[node]);
  if (lx->next_a[node] != JO_LOG_ZERO) intra_word_core(b, j, node + 1, lx->next_a[node]);
  for (k = lx->ac_off[node]; k < lx->ac_off[node + 1]; k++) intra_word_core(b, j, lx->ac_to[k], lx->ac_a[k]);
}

/* save_trellis(), beam.c:2209-2247 */
static int save_trellis(beam *b, const tok *tk, int t)
{
  jamd_trellis_atom *a;
  if (b->natom >= b->atom_cap) { b->overflow = 1; return b->natom - 1; }
  a = &b->atoms[b->natom];
  a->wid = b->lx->stend[tk->node];
  a->backscore = tk->score;
  a->begintime = (short)((tk->last_tre < 0 ? -1 : b->atoms[tk->last_tre].endtime) + 1);
  a->endtime = (short)(t - 1);
  a->last_tre = tk->last_tre;
  a->lscore = tk->last_lscore;
  return b->natom++;
}

/* Entering a word at its root: beam.c:2467-2510 / :2585-2613.  A multipath root has no output, so the
 * token goes one step further along the root's own arcs (self, next, extra) within the same frame. */
static void enter_word(beam *b, int next_node, float tmpsum, int tre, int last_word, float cache)
{
  const jamd_lexicon_desc *lx = b->lx;
  int k;
  if (!b->mp) { propagate_token(b, next_node, tmpsum, tre, last_word, cache); return; }
  if (lx->self_a[next_node] != JO_LOG_ZERO) propagate_token(b, next_node, tmpsum + lx->self_a[next_node], tre, last_word, cache);
  if (lx->next_a[next_node] != JO_LOG_ZERO) propagate_token(b, next_node + 1, tmpsum + lx->next_a[next_node], tre, last_word, cache);
  for (k = lx->ac_off[next_node]; k < lx->ac_off[next_node + 1]; k++)
    propagate_token(b, lx->ac_to[k], tmpsum + lx->ac_a[k], tre, last_word, cache);
}

/* beam_inter_word(), beam.c:2271-2520 (LM_PROB, UNIGRAM_FACTORING).  `li` is the list the source token
 * lives in: the previous frame's (normal mode) or this frame's (multipath, :2781). */
static void inter_word(beam *b, int li, int j, int tre)
{
  const jamd_lexicon_desc *lx = b->lx;
  tok tk = b->tlist[li][b->tindex[li][j]];
  int node = tk.node, sword = lx->stend[node], stid, isoid;
  int last_word = lx->is_transparent[sword] ? tk.last_cword : sword;
  float tmpprob, tmpsum, ngram_score_cache;
  if (sword == lx->tail_silwid) return;
  tmpprob = tk.score + lx->wordend_a[sword];
  if (b->wordend_best_score < tmpprob) {
    b->wordend_best_score = tmpprob; b->wordend_best_node = node;
    b->wordend_best_tre = tre; b->wordend_best_last_cword = tk.last_cword;
  }
  for (stid = lx->startnum - 1; stid >= 0; stid--) {
    if (b->mp && lx->startnode[stid] == lx->word_head[lx->head_silwid]) continue;   /* :2336-2341 */
    isoid = lx->start2isolate[stid];
    if (isoid == -1) continue;
    tmpprob = iw_prob(lx, last_word, stid);      /* iwparray[isoid], keyed by the same word :2323-2327 */
    tmpsum = tk.score;
    tmpsum += lx->wordend_a[sword];
    ngram_score_cache = tmpprob * lx->lm_weight + lx->lm_penalty;
    tmpsum += ngram_score_cache;
    if (lx->is_transparent[sword] && tk.last_cword >= 0 && lx->is_transparent[tk.last_cword])
      tmpsum += lx->lm_penalty_trans;
    enter_word(b, lx->startnode[stid], tmpsum, tre, last_word, ngram_score_cache);
  }
}

/* beam_inter_word(), grammar branch (LM_DFA with category tree, no forward DFA):
 * category-pair test per root (beam.c:2404-2412), word insertion penalty + delayed class
 * penalty of the previous word (:2452-2461) */
static void inter_word_dfa(beam *b, int li, int j, int tre)
{
  const jamd_lexicon_desc *lx = b->lx;
  tok tk = b->tlist[li][b->tindex[li][j]];
  int sword = lx->stend[tk.node], stid;
  int last_word = lx->is_transparent[sword] ? tk.last_cword : sword;
  float tmpsum, ngram_score_cache;
  for (stid = lx->startnum - 1; stid >= 0; stid--) {
    if (!lx->cat_pair[lx->wton[sword] * lx->ncat + lx->wton[lx->start2wid[stid]]]) continue;
    tmpsum = tk.score;
    tmpsum += lx->wordend_a[sword];
    ngram_score_cache = lx->penalty1;
    ngram_score_cache += lx->cprob[last_word];
    tmpsum += ngram_score_cache;
    enter_word(b, lx->startnode[stid], tmpsum, tre, last_word, ngram_score_cache);
  }
}

/* beam_inter_word_factoring(), beam.c:2549-2637 */
static void inter_word_factoring(beam *b)
{
  const jamd_lexicon_desc *lx = b->lx;
  int node = b->wordend_best_node, sword = lx->stend[node], stid, next_node;
  int last_word = lx->is_transparent[sword] ? b->wordend_best_last_cword : sword;
  float tmpsum, ngram_score_cache;
  for (stid = lx->startnum - 1; stid >= 0; stid--) {
    next_node = lx->startnode[stid];
    if (b->mp && next_node == lx->word_head[lx->head_silwid]) continue;      /* :2566-2571 */
    if (lx->start2isolate[stid] != -1) continue;
    ngram_score_cache = lx->fscore[-lx->scid[next_node]] * lx->lm_weight + lx->lm_penalty;
    tmpsum = b->wordend_best_score;
    tmpsum += ngram_score_cache;
    if (lx->is_transparent[sword] && b->wordend_best_last_cword >= 0 &&
        lx->is_transparent[b->wordend_best_last_cword]) tmpsum += lx->lm_penalty_trans;
    if (tmpsum < b->score_pruning_threshold) continue;
    enter_word(b, next_node, tmpsum, b->wordend_best_tre, last_word, ngram_score_cache);
  }
}

/*
 * The whole first pass for one utterance:
 *   get_back_trellis_init()    beam.c:1825 (+ init_nodescore :1552, N-gram branch)
 *   get_back_trellis_proceed() beam.c:2663 for t = 1..T-1 (non-multipath branch)
 *   get_back_trellis_end()     beam.c:3052
 *   find_1pass_result()        beam.c:372 (+ trace_backptr :294)
 * sc is the [T][S] state score matrix (what outprob_state() would return).
 * atoms come out in creation order; last_tre indexes the same array (-1 = bos).
 * Returns 0 = ok, 1 = no sentence-end word survived (search failed), 2 = beam
 * died at frame *died_at (the reference would segment the input there), -1 =
 * atom buffer too small.
 */
/* statistics of the last jo_beam_pass1() call (test sizing: how many tokens a frame creates) */
static int jo_stat_max_tokens = 0; static double jo_stat_sum_tokens = 0.0; static int jo_stat_frames = 0;
int jo_beam_last_max_tokens(void) { return jo_stat_max_tokens; }
double jo_beam_last_mean_tokens(void) { return jo_stat_frames ? jo_stat_sum_tokens / jo_stat_frames : 0.0; }

int jo_beam_pass1(const jamd_lexicon_desc *lx, const float *sc, int T, int S,
                  int beam_width, float score_pruning_width,
                  jamd_trellis_atom *atoms, int atom_cap, int *natom,
                  int *wseq, int wseq_cap, int *wnum, float *pass1_score, int *died_at)
{
  beam B, *b = &B;
  int t, j, i, node, rc = 0;
  memset(b, 0, sizeof(*b));
  b->lx = lx; b->sc = sc; b->S = S; b->atoms = atoms; b->atom_cap = atom_cap;
  b->lmt = lx->lm_type & 0xff; b->mp = (lx->lm_type & JAMD_LM_MULTIPATH) != 0;
  *natom = 0; *wnum = 0; *pass1_score = JO_LOG_ZERO; *died_at = -1;
  jo_stat_max_tokens = 0; jo_stat_sum_tokens = 0.0; jo_stat_frames = 0;
  if (T <= 0) return 1;

  /* get_back_trellis_init */
  b->tn = 0; b->tl = 1;
  b->maxtnum = beam_width * 2 + lx->startnum;             /* malloc_nodes :1871 */
  if (b->maxtnum < 2) b->maxtnum = 2;
  b->expand_step = beam_width > 0 ? beam_width : 1;       /* prepare_nodes :1873 */
  for (i = 0; i < 2; i++) {
    b->tlist[i] = (tok *)malloc(sizeof(tok) * b->maxtnum);
    b->tindex[i] = (int *)malloc(sizeof(int) * b->maxtnum);
  }
  b->token = (int *)malloc(sizeof(int) * lx->nnode);
  for (i = 0; i < lx->nnode; i++) b->token[i] = -1;
  if (b->lmt != JAMD_LM_NGRAM) {                           /* init_nodescore :1669-1757 (grammar), :1762-1788 (word list) */
    int e;
    for (e = 0; e < lx->ninit; e++) {
      int id = create_token(b);
      tok *nw = &b->tlist[b->tn][id];
      node = lx->init_node[e];
      nw->last_tre = -1; nw->last_cword = -1; nw->last_lscore = lx->init_lscore[e];
      nw->score = b->mp ? nw->last_lscore : outprob_style(b, node, -1, 0) + nw->last_lscore;   /* :1733-1737 */
      b->token[node] = id; nw->node = node;
    }
  } else {                                                 /* init_nodescore :1622-1665 */
    int id = create_token(b);
    tok *nw = &b->tlist[b->tn][id];
    node = lx->word_head[lx->head_silwid];
    nw->last_lscore = (lx->scid[node] != 0) ? max_successor_prob(lx, -1, node) : 0.0f;
    nw->last_lscore = nw->last_lscore * lx->lm_weight + lx->lm_penalty;
    nw->last_tre = -1; nw->last_cword = -1;
    nw->score = b->mp ? nw->last_lscore : outprob_style(b, node, -1, 0) + nw->last_lscore;     /* :1657-1663 */
    b->token[node] = id; nw->node = node;
  }
  sort_token_no_order(b, beam_width);
  b->score_pruning_threshold = JO_LOG_ZERO;

  /* get_back_trellis_proceed: frames 1..T-1; a multipath model also runs frame 0 through it
   * (pass1.c:239) and ends with one transition-only call (get_back_trellis_end :3066-3073) */
  for (t = b->mp ? 0 : 1; t < (b->mp ? T + 1 : T); t++) {
    const int final = b->mp && t == T;
    int tl, tn;
    b->tl = b->tn; b->tn = b->tn ? 0 : 1;
    tl = b->tl; tn = b->tn;
    b->wordend_best_score = JO_LOG_ZERO;
    for (j = 0; j < b->tnum[tl]; j++) b->token[b->tlist[tl][j].node] = -1;   /* clear_tokens :1122 */
    if (b->mp) {
      /* :2752-2769 word-internal transitions of every survivor, :2774 beam over the NEW tokens,
       * :2779-2825 word ends among them: trellis word, then cross-word transition within this frame */
      for (j = b->n_start; j <= b->n_end; j++) {
        tok tk = b->tlist[tl][b->tindex[tl][j]];
        if (tk.score <= JO_LOG_ZERO) continue;
        if (tk.score < b->score_pruning_threshold) continue;
        intra_word(b, j);
      }
      sort_token_no_order(b, beam_width);
      for (j = b->n_start; j <= b->n_end; j++) {
        tok tk = b->tlist[tn][b->tindex[tn][j]];
        if (tk.score < b->score_pruning_threshold) continue;
        if (lx->stend[tk.node] >= 0) {
          int tre = save_trellis(b, &tk, t);
          if (final) continue;
          if (b->lmt == JAMD_LM_WORD) continue;
          if (b->lmt == JAMD_LM_DFA) inter_word_dfa(b, tn, j, tre);
          else inter_word(b, tn, j, tre);
        }
      }
    } else {
      for (j = b->n_start; j <= b->n_end; j++) {
        tok tk = b->tlist[tl][b->tindex[tl][j]];
        if (tk.score <= JO_LOG_ZERO) continue;
        if (tk.score < b->score_pruning_threshold) continue;
        intra_word(b, j);
        if (lx->stend[tk.node] >= 0) {
          int tre = save_trellis(b, &tk, t);
          if (b->lmt == JAMD_LM_WORD) continue;               /* isolated words: no cross-word transition :2875 */
          if (b->lmt == JAMD_LM_DFA) inter_word_dfa(b, tl, j, tre);
          else inter_word(b, tl, j, tre);
        }
      }
    }
    if (b->lmt == JAMD_LM_NGRAM && b->wordend_best_score > JO_LOG_ZERO) inter_word_factoring(b);
    b->score_pruning_max = JO_LOG_ZERO;
    if (!final) {
      for (j = 0; j < b->tnum[tn]; j++) {                                     /* :2930-2951 */
        tok *tk = &b->tlist[tn][b->tindex[tn][j]];
        int lw = tk->last_tre < 0 ? -1 : atoms[tk->last_tre].wid;
        if (b->mp && lx->out_kind[tk->node] == JAMD_AS_NONE) continue;        /* non-output node :2935 */
        tk->score += outprob_style(b, tk->node, lw, t);
        if (b->score_pruning_max < tk->score) b->score_pruning_max = tk->score;
      }
    }
    if (score_pruning_width >= 0.0f) b->score_pruning_threshold = b->score_pruning_max - score_pruning_width;
    else b->score_pruning_threshold = JO_LOG_ZERO;
    b->tnum[tl] = 0;                                                           /* clear_tlist */
    if (b->tnum[tn] > jo_stat_max_tokens) jo_stat_max_tokens = b->tnum[tn];
    jo_stat_sum_tokens += b->tnum[tn]; jo_stat_frames++;
    sort_token_no_order(b, beam_width);
    if (b->tnum[tn] == 0) { if (!final) { *died_at = t; rc = 2; } break; }
  }

  if (rc == 0) {
    /* get_back_trellis_end (non-multipath) :3076-3086 */
    if (!b->mp) {
      b->tl = b->tn; b->tn = b->tn ? 0 : 1;
      for (j = b->n_start; j <= b->n_end; j++) {
        tok tk = b->tlist[b->tl][b->tindex[b->tl][j]];
        if (lx->stend[tk.node] >= 0) save_trellis(b, &tk, T);
      }
    }
    /* find_1pass_result :399-431: the tail-silence word ending latest */
    {
      int best = -1, last_time;
      for (last_time = T - 1; last_time >= 0 && best < 0; last_time--) {
        if (b->lmt != JAMD_LM_NGRAM) {
          /* grammar (:433-455), word list (find_1pass_result_word :591-605): the best word on the latest frame that has one; rw[t] is sorted
           * by word id and the test is a strict <, so ties go to the smaller word id */
          float maxscore = JO_LOG_ZERO;
          for (i = 0; i < b->natom; i++)
            if (atoms[i].endtime == last_time &&
                (maxscore < atoms[i].backscore || (maxscore == atoms[i].backscore && best >= 0 && atoms[i].wid < atoms[best].wid))) {
              maxscore = atoms[i].backscore; best = i;
            }
          if (maxscore == JO_LOG_ZERO) best = -1;
          continue;
        }
        for (i = 0; i < b->natom; i++)
          if (atoms[i].endtime == last_time && atoms[i].wid == lx->tail_silwid && atoms[i].backscore > JO_LOG_ZERO) {
            best = i; break;
          }
      }
      if (best < 0) rc = 1;
      else {                                               /* trace_backptr :294-340 */
        int n = 0, k, a = best;
        int *rev = (int *)malloc(sizeof(int) * (b->natom + 1));
        rev[n++] = atoms[a].wid;
        while (atoms[a].begintime > 0) { a = atoms[a].last_tre; rev[n++] = atoms[a].wid; }
        for (k = 0; k < n && k < wseq_cap; k++) wseq[k] = rev[n - 1 - k];
        *wnum = n; *pass1_score = atoms[best].backscore;
        free(rev);
      }
    }
  }
  *natom = b->natom;
  if (b->overflow) rc = -1;
  for (i = 0; i < 2; i++) { free(b->tlist[i]); free(b->tindex[i]); }
  free(b->token);
  return rc;
}
