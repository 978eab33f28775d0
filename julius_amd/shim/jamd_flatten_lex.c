/*
 * jamd_flatten_lex.c -- RecogProcess (WCHMM_INFO + WORD_INFO + NGRAM_INFO +
 * HTK_HMM_INFO context tables) -> jamd_lexicon_desc.  Compiled inside the
 * Julius tree against its own headers; see jamd_flatten.h / INTEGRATION.md.
 *
 * What is walked (reference structures, nothing redefined):
 *   WCHMM_INFO   libjulius/include/julius/wchmm.h:211-278
 *   A_CELL2      wchmm.h:162-172   (extra arcs; list order kept)
 *   RC_INFO / LRC_INFO / ACOUSTIC_SPEC   wchmm.h:55-98
 *   WORD_INFO    libsent/include/sent/vocabulary.h:53-86
 *   NGRAM_INFO / NGRAM_TUPLE_INFO        libsent/include/sent/ngram2.h:137-188
 * The cross-word context table reproduces, for every (word-head node, left
 * context phone) pair, exactly the choice outprob_style() makes at run time
 * (libjulius/src/outprob_style.c:385-486) by calling the same lookup helpers
 * (get_left_context_HMM, lcdset_lookup_by_hmmname).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#define JAMD_WITH_LIBJULIUS 1
#include "jamd_flatten.h"

#define NEW(T, n) ((T *)calloc((size_t)((n) > 0 ? (n) : 1), sizeof(T)))

/* ---- small pointer -> id map ------------------------------------------- */
typedef struct { const void **key; int *val; size_t cap; int n; } pmap;
static void pmap_init(pmap *m, size_t want) {
  m->cap = 64; while (m->cap < 2 * want + 16) m->cap <<= 1;
  m->key = (const void **)calloc(m->cap, sizeof(void *));
  m->val = (int *)malloc(m->cap * sizeof(int));
  m->n = 0;
}
static void pmap_grow(pmap *m) {
  pmap o = *m; size_t i;
  m->cap = o.cap << 1;
  m->key = (const void **)calloc(m->cap, sizeof(void *));
  m->val = (int *)malloc(m->cap * sizeof(int));
  for (i = 0; i < o.cap; i++) if (o.key[i]) {
    size_t x = (size_t)o.key[i]; x ^= x >> 17; x *= (size_t)0x9E3779B97F4A7C15ull; x ^= x >> 29;
    x &= m->cap - 1;
    while (m->key[x]) x = (x + 1) & (m->cap - 1);
    m->key[x] = o.key[i]; m->val[x] = o.val[i];
  }
  free(o.key); free(o.val);
}
/* returns existing id, or assigns the next id (and sets *isnew) */
static int pmap_id(pmap *m, const void *k, int *isnew) {
  size_t x;
  if ((size_t)m->n * 2 + 2 > m->cap) pmap_grow(m);
  x = (size_t)k; x ^= x >> 17; x *= (size_t)0x9E3779B97F4A7C15ull; x ^= x >> 29; x &= m->cap - 1;
  while (m->key[x] != NULL && m->key[x] != k) x = (x + 1) & (m->cap - 1);
  if (m->key[x] == NULL) { m->key[x] = k; m->val[x] = m->n++; if (isnew) *isnew = 1; return m->val[x]; }
  if (isnew) *isnew = 0;
  return m->val[x];
}
static void pmap_free(pmap *m) { free(m->key); free(m->val); }

/* ---- growing int array -------------------------------------------------- */
typedef struct { int *v; int n, cap; } ivec;
static void ivec_push(ivec *a, int x) {
  if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 256; a->v = (int *)realloc(a->v, sizeof(int) * a->cap); }
  a->v[a->n++] = x;
}

typedef struct {
  pmap sets;         /* CD_State_Set* -> set id */
  ivec set_off, set_states;
} setreg;

static int set_id(setreg *sr, CD_State_Set *cs)
{
  int isnew, id = pmap_id(&sr->sets, cs, &isnew), k;
  if (isnew) {
    for (k = 0; k < cs->num; k++) ivec_push(&sr->set_states, cs->s[k]->id);
    ivec_push(&sr->set_off, sr->set_states.n);
  }
  return id;
}

/* Multipath lexicons (model-skip transitions; non-emitting word-begin and word-end nodes) flatten to
 * the same descriptor with lm_type | JAMD_LM_MULTIPATH, JAMD_AS_NONE on the non-emitting nodes and
 * word_head[] = wchmm->wordbegin[].  The device first pass does not take them yet
 * (jamd_lexicon_create() refuses the flag), so only the explicit entry below produces them. */
static int g_allow_multipath = 0;
int jamd_flatten_lexicon_multipath(RecogProcess *r, jamd_flat_lexicon *out)
{
  int rc;
  g_allow_multipath = 1;
  rc = jamd_flatten_lexicon(r, out);
  g_allow_multipath = 0;
  return rc;
}

int jamd_flatten_lexicon(RecogProcess *r, jamd_flat_lexicon *out)
{
  WCHMM_INFO *wchmm = r->wchmm;
  WORD_INFO *winfo = wchmm->winfo;
  HTK_HMM_INFO *hmminfo = wchmm->hmminfo;
  NGRAM_INFO *ng = wchmm->ngram;
  jamd_lexicon_desc *d = &out->desc;
  int n = wchmm->n, W = winfo->num, i, k, w;
  setreg sr;
  char **lcname = NULL; int nlc = 0;
  int *word_lc;
  pmap rows; ivec rowkey_kind; ivec lc_tab;
  typedef struct { HMM_Logical *hmm; short loc; unsigned char kind; int cat; } rowkey;
  rowkey *rk = NULL; int nrk = 0, caprk = 0;
  char rbuf[MAX_HMMNAME_LEN], cbuf[MAX_HMMNAME_LEN];

  memset(out, 0, sizeof(*out));
  const int dfa_mode = (r->lmtype == LM_DFA);
  const int word_mode = (dfa_mode && r->lmvar == LM_DFA_WORD);      /* isolated word recognition (-w) */
  const int multipath = hmminfo->multipath ? 1 : 0;
  if (multipath && !g_allow_multipath) return JAMD_EINVAL;
  if (word_mode) {
    if (!wchmm->category_tree) return JAMD_EINVAL;
  } else if (dfa_mode) {              /* grammar: per-category trees (with or without a forward DFA) */
    if (r->lmvar != LM_DFA_GRAMMAR || !wchmm->category_tree || wchmm->dfa == NULL)
      return JAMD_EINVAL;
  } else {
    if (r->lmtype != LM_PROB || ng == NULL) return JAMD_EINVAL;
    if (wchmm->category_tree) return JAMD_EINVAL;
    if (r->lmvar == LM_NGRAM_USER) return JAMD_EINVAL;
  }

  memset(&sr, 0, sizeof(sr)); pmap_init(&sr.sets, 1024); ivec_push(&sr.set_off, 0);
  memset(&rowkey_kind, 0, sizeof(rowkey_kind)); memset(&lc_tab, 0, sizeof(lc_tab));
  pmap_init(&rows, 1024);

  /* ---- left-context classes: base phone of each word's last phone -------- */
  word_lc = NEW(int, W);
  for (w = 0; w < W; w++) {
    center_name(winfo->wseq[w][winfo->wlen[w] - 1]->name, cbuf);
    for (k = 0; k < nlc; k++) if (strcmp(lcname[k], cbuf) == 0) break;
    if (k == nlc) { lcname = (char **)realloc(lcname, sizeof(char *) * (nlc + 1)); lcname[nlc++] = strdup(cbuf); }
    word_lc[w] = k;
  }

  /* ---- nodes ---------------------------------------------------------------- */
  out->self_a = NEW(float, n); out->next_a = NEW(float, n); out->ac_off = NEW(int, n + 1);
  out->stend = NEW(int, n); out->scid = NEW(int, n); out->out_kind = NEW(unsigned char, n); out->out_id = NEW(int, n);
  {
    ivec ato; memset(&ato, 0, sizeof(ato));
    float *aa = NULL; int acap = 0, an = 0;
    for (i = 0; i < n; i++) {
      A_CELL2 *ac;
      out->self_a[i] = wchmm->self_a[i];
      out->next_a[i] = wchmm->next_a[i];
      out->ac_off[i] = an;
      for (ac = wchmm->ac[i]; ac; ac = ac->next) {
        for (k = 0; k < ac->n; k++) {
          if (an == acap) { acap = acap ? acap * 2 : 1024; aa = (float *)realloc(aa, sizeof(float) * acap); }
          ivec_push(&ato, ac->arc[k]); aa[an++] = ac->a[k];
        }
      }
      out->stend[i] = (wchmm->stend[i] == WORD_INVALID) ? -1 : (int)wchmm->stend[i];
      out->scid[i] = dfa_mode ? 0 : wchmm->state[i].scid;   /* per-category trees carry no factoring data */
    }
    out->ac_off[n] = an;
    out->ac_to = ato.v ? ato.v : NEW(int, 1);
    out->ac_a = aa ? aa : NEW(float, 1);
  }

  /* ---- acoustic spec of every node (outprob_style.c:376-486) ---------------- */
  for (i = 0; i < n; i++) {
    unsigned char kind = wchmm->ccd_flag ? wchmm->outstyle[i] : AS_STATE;
    if (wchmm->state[i].out.state == NULL) {                   /* non-emitting: multipath only (beam.c:2935) */
      if (!multipath) { free(word_lc); return JAMD_EINVAL; }
      out->out_kind[i] = JAMD_AS_NONE; out->out_id[i] = -1;
      continue;
    }
    switch (kind) {
    case AS_STATE:
      out->out_kind[i] = JAMD_AS_STATE; out->out_id[i] = wchmm->state[i].out.state->id; break;
    case AS_LSET:
      out->out_kind[i] = JAMD_AS_LSET; out->out_id[i] = set_id(&sr, wchmm->state[i].out.lset); break;
    case AS_RSET: case AS_LRSET: {
      HMM_Logical *base = (kind == AS_RSET) ? wchmm->state[i].out.rset->hmm : wchmm->state[i].out.lrset->hmm;
      short loc = (kind == AS_RSET) ? wchmm->state[i].out.rset->state_loc : wchmm->state[i].out.lrset->state_loc;
      /* with per-category trees a one-phone word's state set also depends on its category
       * (lcdset_lookup_with_category(), outprob_style.c:144, :452-458) */
      const int cat = (kind == AS_LRSET && wchmm->category_tree) ? (int)wchmm->state[i].out.lrset->category : -1;
      int row = -1;
      for (k = 0; k < nrk; k++) if (rk[k].hmm == base && rk[k].loc == loc && rk[k].kind == kind && rk[k].cat == cat) { row = k; break; }
      if (row < 0) {
        int c;
        if (nrk == caprk) { caprk = caprk ? caprk * 2 : 256; rk = (rowkey *)realloc(rk, sizeof(rowkey) * caprk); }
        rk[nrk].hmm = base; rk[nrk].loc = loc; rk[nrk].kind = kind; rk[nrk].cat = cat; row = nrk++;
        for (c = 0; c <= nlc; c++) {          /* column nlc: last_wid == WORD_INVALID */
          HMM_Logical *rhmm = base, *ohmm;
          int ent;
          if (kind == AS_RSET) {                                   /* outprob_style.c:390-428 */
            if (c < nlc && (ohmm = get_left_context_HMM(base, lcname[c], hmminfo)) != NULL) rhmm = ohmm;
            if (rhmm->is_pseudo) ent = ~set_id(&sr, &(rhmm->body.pseudo->stateset[loc]));
            else ent = rhmm->body.defined->s[loc]->id;
          } else {                                                 /* outprob_style.c:440-478 */
            CD_Set *lcd;
            if (wchmm->category_tree) {
              if (c < nlc && (ohmm = get_left_context_HMM(base, lcname[c], hmminfo)) != NULL)
                lcd = lcdset_lookup_with_category(wchmm, ohmm, (WORD_ID)cat);
              else
                lcd = lcdset_lookup_with_category(wchmm, base, (WORD_ID)cat);
            } else {
              strcpy(rbuf, base->name);
              if (c < nlc) add_left_context(rbuf, lcname[c]);
              lcd = lcdset_lookup_by_hmmname(hmminfo, rbuf);
            }
            if (lcd != NULL) ent = ~set_id(&sr, &(lcd->stateset[loc]));
            else if (base->is_pseudo) ent = ~set_id(&sr, &(base->body.pseudo->stateset[loc]));
            else ent = base->body.defined->s[loc]->id;
          }
          ivec_push(&lc_tab, ent);
        }
      }
      out->out_kind[i] = (kind == AS_RSET) ? JAMD_AS_RSET : JAMD_AS_LRSET;
      out->out_id[i] = row;
      break; }
    default:
      free(word_lc); return JAMD_EINVAL;
    }
  }
  free(rk); pmap_free(&rows);

  /* ---- roots, words ------------------------------------------------------------ */
  out->startnode = NEW(int, wchmm->startnum); out->start2isolate = NEW(int, wchmm->startnum);
  for (i = 0; i < wchmm->startnum; i++) {
    out->startnode[i] = wchmm->startnode[i];
    out->start2isolate[i] = (!dfa_mode && wchmm->start2isolate) ? wchmm->start2isolate[i] : -1;
  }
  out->wordend_a = NEW(float, W); out->wton = NEW(int, W); out->cprob = NEW(float, W);
  out->is_transparent = NEW(unsigned char, W); out->word_head = NEW(int, W);
  for (w = 0; w < W; w++) {
    out->wordend_a[w] = multipath ? 0.0f : wchmm->wordend_a[w]; out->wton[w] = winfo->wton[w]; out->cprob[w] = winfo->cprob[w];
    out->is_transparent[w] = winfo->is_transparent[w] ? 1 : 0;
    out->word_head[w] = multipath ? wchmm->wordbegin[w] : wchmm->offset[w][0];     /* beam.c:1635-1639 */
  }
  out->word_lc = word_lc;

  /* ---- factoring ------------------------------------------------------------------ */
  if (dfa_mode) {                       /* fsnum / scnum are never set with per-category trees (wchmm.c:1938) */
    out->fscore = NEW(float, 1); out->scword = NEW(int, 1);
  } else {
    out->fscore = NEW(float, wchmm->fsnum); for (i = 0; i < wchmm->fsnum; i++) out->fscore[i] = wchmm->fscore[i];
    out->scword = NEW(int, wchmm->scnum);
    for (i = 1; i < wchmm->scnum; i++) out->scword[i] = wchmm->scword[i];
  }

  /* ---- forward 2-gram (bi_prob_func_set(), ngram_access.c:449-466) --------------------- */
  if (!dfa_mode) {
    int V = ng->max_word_num;
    NGRAM_TUPLE_INFO *t2 = &(ng->d[1]);
    const float *bo, *bp;
    if (t2->is24bit || t2->bgn == NULL) { free(word_lc); return JAMD_EINVAL; }
    if (ng->bigram_index_reversed) { d->ng_mode = JAMD_NG_ADDITIONAL_OLD; bo = ng->bo_wt_1; bp = ng->p_2; }
    else if (ng->dir == DIR_LR) { d->ng_mode = JAMD_NG_NORMAL; bo = ng->d[0].bo_wt; bp = t2->prob; }
    else if (ng->bo_wt_1 != NULL) { d->ng_mode = JAMD_NG_ADDITIONAL; bo = ng->bo_wt_1; bp = ng->p_2; }
    else { d->ng_mode = JAMD_NG_COMPUTE; bo = ng->d[0].bo_wt; bp = t2->prob; }
    out->ng_uni_prob = NEW(float, V); out->ng_uni_bo = NEW(float, V);
    out->ng_bi_bgn = NEW(int, V); out->ng_bi_num = NEW(int, V);
    for (i = 0; i < V; i++) {
      out->ng_uni_prob[i] = ng->d[0].prob[i]; out->ng_uni_bo[i] = bo[i];
      out->ng_bi_bgn[i] = (t2->bgn[i] == NNID_INVALID) ? -1 : (int)t2->bgn[i];
      out->ng_bi_num[i] = t2->num[i];
    }
    out->ng_bi_wid = NEW(int, t2->totalnum); out->ng_bi_prob = NEW(float, t2->totalnum);
    for (i = 0; i < (int)t2->totalnum; i++) { out->ng_bi_wid[i] = t2->nnid2wid[i]; out->ng_bi_prob[i] = bp[i]; }
    d->ng_nword = V; d->ng_nbigram = t2->totalnum;
    d->ng_unk_id = (int)ng->unk_id; d->ng_unk_num_log = ng->unk_num_log;
  } else {
    /* ---- grammar: category pairs, roots' categories, initial tokens (beam.c:1669-1757) ---- */
    DFA_INFO *dfa = wchmm->dfa;
    MULTIGRAM *m;
    int C = word_mode ? 1 : dfa->term_num, c1, c2, t, iw, ninit = 0;
    int *seen = NEW(int, n);
    out->ng_uni_prob = NEW(float, 1); out->ng_uni_bo = NEW(float, 1); out->ng_bi_bgn = NEW(int, 1);
    out->ng_bi_num = NEW(int, 1); out->ng_bi_wid = NEW(int, 1); out->ng_bi_prob = NEW(float, 1);
    out->cat_pair = NEW(unsigned char, C * C);
    if (!word_mode)
      for (c1 = 0; c1 < C; c1++) for (c2 = 0; c2 < C; c2++) out->cat_pair[c1 * C + c2] = dfa_cp(dfa, c1, c2) ? 1 : 0;
    out->start2wid = NEW(int, wchmm->startnum);
    for (i = 0; i < wchmm->startnum; i++) out->start2wid[i] = wchmm->start2wid[i];
    out->init_node = NEW(int, W); out->init_lscore = NEW(float, W);
    if (!word_mode && wchmm->dfa_forward != NULL) {
      /* the forward DFA as CSR, arcs in list order (the reference takes the first arc whose label matches, beam.c:1741, :2415) */
      DFA_INFO *fw = wchmm->dfa_forward;
      DFA_ARC *ac;
      int s2, na = 0;
      for (s2 = 0; s2 < fw->state_num; s2++) for (ac = fw->st[s2].arc; ac; ac = ac->next) na++;
      out->fwd_off = NEW(int, fw->state_num + 1); out->fwd_label = NEW(int, na + 1); out->fwd_to = NEW(int, na + 1);
      out->init_to_state = NEW(int, W);
      na = 0;
      for (s2 = 0; s2 < fw->state_num; s2++) {
        out->fwd_off[s2] = na;
        for (ac = fw->st[s2].arc; ac; ac = ac->next) { out->fwd_label[na] = ac->label; out->fwd_to[na] = ac->to_state; na++; }
      }
      out->fwd_off[fw->state_num] = na;
      d->nfwd = fw->state_num;
    }
    for (m = r->lm->grammars; m; m = m->next) {
      if (!m->active) continue;
      if (word_mode) {                                 /* every word of the active lists, beam.c:1762-1788 */
        for (iw = m->word_begin; iw < m->word_begin + m->winfo->num; iw++) {
          int node = multipath ? wchmm->wordbegin[iw] : wchmm->offset[iw][0];
          if (seen[node]) continue;
          seen[node] = 1;
          out->init_node[ninit] = node;
          out->init_lscore[ninit] = 0.0f;
          ninit++;
        }
        continue;
      }
      for (t = m->cate_begin; t < m->cate_begin + m->dfa->term_num; t++) {
        if (dfa_cp_begin(dfa, t) != TRUE) continue;
        for (iw = 0; iw < dfa->term.wnum[t]; iw++) {
          int wid = dfa->term.tw[t][iw], node = multipath ? wchmm->wordbegin[wid] : wchmm->offset[wid][0];
          if (seen[node]) continue;                    /* node_exist_token(), beam.c:1717 */
          seen[node] = 1;
          out->init_node[ninit] = node;
          out->init_lscore[ninit] = r->config->lmp.penalty1 + winfo->cprob[wid];   /* beam.c:1724-1727 */
          if (out->init_to_state != NULL) {            /* beam.c:1739-1747: from boslist[gram].dfa_state = m->state_begin (:1694) */
            int a2, ts = -1;
            if (m->state_begin >= 0 && m->state_begin < d->nfwd)
              for (a2 = out->fwd_off[m->state_begin]; a2 < out->fwd_off[m->state_begin + 1]; a2++)
                if (out->fwd_label[a2] == t) { ts = out->fwd_to[a2]; break; }
            out->init_to_state[ninit] = ts;
          }
          ninit++;
        }
      }
    }
    free(seen);
    d->lm_type = word_mode ? JAMD_LM_WORD : JAMD_LM_DFA; d->ncat = C; d->ninit = ninit; d->penalty1 = r->config->lmp.penalty1;
    d->cat_pair = out->cat_pair; d->start2wid = out->start2wid; d->init_node = out->init_node; d->init_lscore = out->init_lscore;
    d->fwd_off = out->fwd_off; d->fwd_label = out->fwd_label; d->fwd_to = out->fwd_to; d->init_to_state = out->init_to_state;
    d->ng_unk_id = -1;
  }

  /* ---- descriptor ------------------------------------------------------------------------- */
  out->lc_tab = lc_tab.v ? lc_tab.v : NEW(int, 1);
  out->set_off = sr.set_off.v; out->set_states = sr.set_states.v ? sr.set_states.v : NEW(int, 1);
  d->nnode = n; d->nword = W; d->startnum = wchmm->startnum; d->isolatenum = dfa_mode ? 0 : wchmm->isolatenum;
  d->self_a = out->self_a; d->next_a = out->next_a; d->ac_off = out->ac_off; d->ac_to = out->ac_to; d->ac_a = out->ac_a;
  d->stend = out->stend; d->scid = out->scid; d->out_kind = out->out_kind; d->out_id = out->out_id;
  d->nlc = nlc; d->nlcrow = nrk; d->lc_tab = out->lc_tab; d->word_lc = out->word_lc;
  d->nset = sr.sets.n; d->set_off = out->set_off; d->set_states = out->set_states;
  d->cdset_method = (hmminfo->cdset_method == IWCD_MAX) ? JAMD_IWCD_MAX :
                    (hmminfo->cdset_method == IWCD_AVG) ? JAMD_IWCD_AVG : JAMD_IWCD_NBEST;
  d->cdmax_num = hmminfo->cdmax_num;
  d->startnode = out->startnode; d->start2isolate = out->start2isolate;
  d->wordend_a = out->wordend_a; d->wton = out->wton; d->cprob = out->cprob;
  d->is_transparent = out->is_transparent; d->word_head = out->word_head;
  d->head_silwid = (winfo->head_silwid == WORD_INVALID) ? -1 : (int)winfo->head_silwid;
  d->tail_silwid = (winfo->tail_silwid == WORD_INVALID) ? -1 : (int)winfo->tail_silwid;
  d->nfscore = dfa_mode ? 1 : wchmm->fsnum; d->nscword = dfa_mode ? 1 : wchmm->scnum; d->fscore = out->fscore; d->scword = out->scword;
  d->ng_uni_prob = out->ng_uni_prob; d->ng_uni_bo = out->ng_uni_bo; d->ng_bi_bgn = out->ng_bi_bgn;
  d->ng_bi_num = out->ng_bi_num; d->ng_bi_wid = out->ng_bi_wid; d->ng_bi_prob = out->ng_bi_prob;
  if (multipath) d->lm_type |= JAMD_LM_MULTIPATH;
  d->lm_weight = r->config->lmp.lm_weight; d->lm_penalty = r->config->lmp.lm_penalty;
  d->lm_penalty_trans = r->config->lmp.lm_penalty_trans;
  for (k = 0; k < nlc; k++) free(lcname[k]);
  free(lcname); pmap_free(&sr.sets);
  return JAMD_OK;
}

void jamd_flat_lexicon_free(jamd_flat_lexicon *f)
{
  free(f->self_a); free(f->next_a); free(f->ac_off); free(f->ac_to); free(f->ac_a); free(f->stend);
  free(f->scid); free(f->out_kind); free(f->out_id); free(f->lc_tab); free(f->word_lc); free(f->set_off);
  free(f->set_states); free(f->startnode); free(f->start2isolate); free(f->wordend_a); free(f->wton);
  free(f->cprob); free(f->is_transparent); free(f->word_head); free(f->fscore); free(f->scword);
  free(f->ng_uni_prob); free(f->ng_uni_bo); free(f->ng_bi_bgn); free(f->ng_bi_num); free(f->ng_bi_wid);
  free(f->ng_bi_prob); free(f->cat_pair); free(f->start2wid); free(f->init_node); free(f->init_lscore);
  free(f->fwd_off); free(f->fwd_label); free(f->fwd_to); free(f->init_to_state);
  memset(f, 0, sizeof(*f));
}

/* ---- blob writer -------------------------------------------------------------------
 * "JAMDLEX1", int32 nrec, then per record: char name[24], int32 dtype (0 = int32,
 * 1 = float32, 2 = uint8), int32 count, payload padded to a multiple of 4 bytes.
 * Scalars travel as the records "ints" and "floats" (order below).  Host byte order. */
static int put_rec(FILE *f, const char *name, int dtype, int count, const void *data)
{
  char nm[24]; static const char zero[4] = {0, 0, 0, 0};
  size_t bytes = (size_t)count * (dtype == 2 ? 1 : 4), pad = (4 - (bytes & 3)) & 3;
  memset(nm, 0, sizeof(nm)); strncpy(nm, name, sizeof(nm) - 1);
  if (fwrite(nm, 1, 24, f) != 24) return -1;
  if (fwrite(&dtype, 4, 1, f) != 1 || fwrite(&count, 4, 1, f) != 1) return -1;
  if (bytes && fwrite(data, 1, bytes, f) != bytes) return -1;
  if (pad && fwrite(zero, 1, pad, f) != pad) return -1;
  return 0;
}

/* Appends the N-gram vocabulary (every name followed by NUL, in N-gram id order) to a lexicon file as the record
 * "ng_wname": jamd_lexicon_load_ngram() / jamd_bingram_check() refuse a binary N-gram over another vocabulary (the
 * tree's word -> N-gram ids would not fit).  The record count in the header is bumped. */
int jamd_lexicon_append_ngram_names(const char *path, const NGRAM_INFO *ng)
{
  FILE *f;
  int nrec = 0, i;
  size_t len = 0, at = 0;
  char *buf;
  if (ng == NULL || ng->wname == NULL) return JAMD_OK;              /* a grammar: nothing to record */
  for (i = 0; i < (int)ng->max_word_num; i++) len += strlen(ng->wname[i]) + 1;
  if ((buf = (char *)malloc(len + 4)) == NULL) return JAMD_ENOMEM;
  for (i = 0; i < (int)ng->max_word_num; i++) { size_t n = strlen(ng->wname[i]) + 1; memcpy(buf + at, ng->wname[i], n); at += n; }
  if ((f = fopen(path, "r+b")) == NULL) { free(buf); return JAMD_EINVAL; }
  if (fseek(f, 8, SEEK_SET) != 0 || fread(&nrec, 4, 1, f) != 1) { fclose(f); free(buf); return JAMD_EINVAL; }
  nrec++;
  if (fseek(f, 8, SEEK_SET) != 0 || fwrite(&nrec, 4, 1, f) != 1 || fseek(f, 0, SEEK_END) != 0 ||
      put_rec(f, "ng_wname", 2, (int)len, buf) != 0) { fclose(f); free(buf); return JAMD_EINVAL; }
  free(buf);
  return fclose(f) == 0 ? JAMD_OK : JAMD_EINVAL;
}

/* Appends "sep_wnum" (-sepnum: how many of the most frequent words wchmm.c keeps out of the tree, wchmm.c:1782, :1882) so
 * that jamd_lexicon_load_ngram() can tell whether a retrained N-gram would have produced ANOTHER tree (then it refuses)
 * before it recomputes the 1-gram factoring values of this one. */
int jamd_lexicon_append_separation(const char *path, int separate_wnum)
{
  FILE *f = fopen(path, "r+b");
  int nrec = 0;
  if (f == NULL) return JAMD_EINVAL;
  if (fseek(f, 8, SEEK_SET) != 0 || fread(&nrec, 4, 1, f) != 1) { fclose(f); return JAMD_EINVAL; }
  nrec++;
  if (fseek(f, 8, SEEK_SET) != 0 || fwrite(&nrec, 4, 1, f) != 1 || fseek(f, 0, SEEK_END) != 0 ||
      put_rec(f, "sep_wnum", 0, 1, &separate_wnum) != 0) { fclose(f); return JAMD_EINVAL; }
  return fclose(f) == 0 ? JAMD_OK : JAMD_EINVAL;
}

int jamd_lexicon_save(const jamd_lexicon_desc *d, const char *path)
{
  FILE *f = fopen(path, "wb");
  int nrec = 30 + ((d->lm_type & 0xff) != JAMD_LM_NGRAM ? 4 : 0) + (d->nfwd > 0 ? 4 : 0), rc = 0;
  int ints[22]; float floats[5];
  if (f == NULL) return JAMD_EINVAL;
  ints[0] = d->nnode; ints[1] = d->nword; ints[2] = d->startnum; ints[3] = d->isolatenum;
  ints[4] = d->nlc; ints[5] = d->nlcrow; ints[6] = d->nset; ints[7] = d->cdset_method; ints[8] = d->cdmax_num;
  ints[9] = d->head_silwid; ints[10] = d->tail_silwid; ints[11] = d->nfscore; ints[12] = d->nscword;
  ints[13] = d->ng_mode; ints[14] = d->ng_nword; ints[15] = d->ng_nbigram; ints[16] = d->ng_unk_id; ints[17] = 0;
  ints[18] = d->lm_type; ints[19] = d->ncat; ints[20] = d->ninit; ints[21] = d->nfwd; floats[4] = d->penalty1;
  floats[0] = d->ng_unk_num_log; floats[1] = d->lm_weight; floats[2] = d->lm_penalty; floats[3] = d->lm_penalty_trans;
  fwrite("JAMDLEX1", 1, 8, f); fwrite(&nrec, 4, 1, f);
#define I32(nm, p, n) rc |= put_rec(f, nm, 0, (n), (p))
#define F32(nm, p, n) rc |= put_rec(f, nm, 1, (n), (p))
#define U8(nm, p, n)  rc |= put_rec(f, nm, 2, (n), (p))
  I32("ints", ints, 22); F32("floats", floats, 5);
  F32("self_a", d->self_a, d->nnode); F32("next_a", d->next_a, d->nnode);
  I32("ac_off", d->ac_off, d->nnode + 1); I32("ac_to", d->ac_to, d->ac_off[d->nnode]); F32("ac_a", d->ac_a, d->ac_off[d->nnode]);
  I32("stend", d->stend, d->nnode); I32("scid", d->scid, d->nnode);
  U8("out_kind", d->out_kind, d->nnode); I32("out_id", d->out_id, d->nnode);
  I32("lc_tab", d->lc_tab, d->nlcrow * (d->nlc + 1)); I32("word_lc", d->word_lc, d->nword);
  I32("set_off", d->set_off, d->nset + 1); I32("set_states", d->set_states, d->set_off[d->nset]);
  I32("startnode", d->startnode, d->startnum); I32("start2isolate", d->start2isolate, d->startnum);
  F32("wordend_a", d->wordend_a, d->nword); I32("wton", d->wton, d->nword); F32("cprob", d->cprob, d->nword);
  U8("is_transparent", d->is_transparent, d->nword); I32("word_head", d->word_head, d->nword);
  F32("fscore", d->fscore, d->nfscore); I32("scword", d->scword, d->nscword);
  F32("ng_uni_prob", d->ng_uni_prob, d->ng_nword); F32("ng_uni_bo", d->ng_uni_bo, d->ng_nword);
  I32("ng_bi_bgn", d->ng_bi_bgn, d->ng_nword); I32("ng_bi_num", d->ng_bi_num, d->ng_nword);
  I32("ng_bi_wid", d->ng_bi_wid, d->ng_nbigram); F32("ng_bi_prob", d->ng_bi_prob, d->ng_nbigram);
  if ((d->lm_type & 0xff) != JAMD_LM_NGRAM) {
    U8("cat_pair", d->cat_pair, d->ncat * d->ncat); I32("start2wid", d->start2wid, d->startnum);
    I32("init_node", d->init_node, d->ninit); F32("init_lscore", d->init_lscore, d->ninit);
    if (d->nfwd > 0) {
      I32("fwd_off", d->fwd_off, d->nfwd + 1); I32("fwd_label", d->fwd_label, d->fwd_off[d->nfwd]);
      I32("fwd_to", d->fwd_to, d->fwd_off[d->nfwd]); I32("init_to_state", d->init_to_state, d->ninit);
    }
  }
#undef I32
#undef F32
#undef U8
  if (fclose(f) != 0) rc = -1;
  return rc ? JAMD_EINVAL : JAMD_OK;
}
