// beam.hip -- first-pass token passing over the tree lexicon (K6) for gfx950.
//
// Replaces, for a BATCH of utterances in one launch, the reference's
//   get_back_trellis_init()     libjulius/src/beam.c:1825  (+ init_nodescore :1552)
//   get_back_trellis_proceed()  beam.c:2663   (non-multipath branch :2832-2900)
//     beam_intra_word()/_core() beam.c:2154 / :2004
//     save_trellis()            beam.c:2209
//     beam_inter_word()         beam.c:2271
//     beam_inter_word_factoring beam.c:2549
//     sort_token_no_order()     beam.c:1492   (rank pruning)
//   get_back_trellis_end()      beam.c:3052
//   find_1pass_result()         beam.c:372    (+ trace_backptr :294)
// and their callees outprob_style() (outprob_style.c:354), max_successor_prob()
// / max_successor_prob_iw() (factoring_sub.c:942 / :1049) and the 2-gram access
// functions (libsent/src/ngram/ngram_access.c:225-403).
//
// Execution model: ONE WORKGROUP PER UTTERANCE, persistent over all frames of
// that utterance (the trellis is serial in t; utterances are independent, so a
// batch fills the chip).  Per frame:
//   A  every surviving token pushes its intra-word candidates, emits a trellis
//      atom if it sits on a word end and records itself in the word-end list;
//   B  word ends x isolated roots (2-gram) and best word end x shared roots
//      (1-gram factoring) push the cross-word candidates;
//   C  one thread per touched node takes the winner, rebuilds its payload from
//      the winning candidate id, adds the acoustic score, tracks the frame max;
//   D  rank pruning: radix select of the beam_width-th score + compaction.
// "push" is a 64-bit atomicMax on the node's Viterbi cell = (order-preserving score bits,
// candidate id): the Viterbi max of propagate_token() (beam.c:1945) without any
// ordering between candidates.  The cells of the frame being built live in an LDS hash
// table keyed by node (struct Cells; nodekey[] in global memory is its overflow).  The
// thread that claims a cell registers the node in the frame's touched list, step C
// empties the cell again, so both tables are clean after every frame without a clearing
// sweep (what clear_tokens(), beam.c:1122, does on the CPU).
// Grammar (per-category trees) and isolated-word recognition run through the same kernel
// (lx.lm_type): initial tokens enter through steps C and D of a pseudo frame 0, step B
// becomes word ends x all roots with the category-pair test, or nothing at all.
//
// Determinism / parity: every float is produced by the same sequence of fp32
// operations as the reference (this file is compiled with -ffp-contract=off).
// The only freedom is the visiting order, which matters only when two candidates
// for the same node, two word ends, or two tokens at the rank cut have EXACTLY
// equal scores; every such event is counted in jamd_pass1_result.ties.  With
// ties == 0 the trellis is the reference's trellis bit for bit.
#include "jamd_device.h"
#include <type_traits>
#include <algorithm>

#include "beam_common.h"
#include "beam_exact.h"

namespace {
using namespace jamdb;
// candidate ids (low 32 bits of a node key) name the SOURCE of the transition, in
// terms that do not depend on any scheduling order, so that (score, id) is a
// canonical total order and the result is deterministic:
//   intra-word     bit31 = 0            [30:0] = source node
//   isolated root  bits[31:30] = 10     [29:0] = the word that ended (its end node is unique);
//                                       with a grammar every root is entered this way
//   shared root    bits[31:30] = 11     (the source is the frame's best word end);
//                                       with a grammar: [29:0] = index of an initial token
// The destination is the address of the key, so the arc is implied.
// Returns the previous key when it holds the SAME score as this candidate (an
// exact tie), else 0.
// Where the Viterbi cells of the frame being built live.  A frame touches a few thousand of the
// lexicon's 10^5..10^6 nodes; with hundreds of utterances in flight the direct-indexed nodekey[]
// tables (2 MB each) fall out of every cache and each push becomes a random DRAM read-modify-write.
// The cells therefore live in an LDS hash table keyed by node (open addressing, claimed with a CAS
// on the node word); a node whose probe window is full overflows to nodekey[] -- occupancy only
// grows within a frame, so every candidate of a node resolves to the same place.
struct Cells {
  unsigned char *ub;             // this utterance's slice: nodekey[] (overflow, and everything when nslot == 0), touched[]
  unsigned o_nodekey, o_touched;
  unsigned long long *lkey;      // [nslot] LDS cells (0 = empty)
  int *lnode;                    // [nslot] owning node (-1 = free)
  int nslot, shift;              // nslot = 1 << (32 - shift)
};
constexpr int kCellProbes = 24;

__device__ __forceinline__ unsigned long long push(Shared &sh, const Cells &cl, int node, float score, unsigned id) {
  if (score <= JAMD_LOG_ZERO) return 0ull;                    // propagate_token() :1951
  const unsigned long long key = ((unsigned long long)ord(score) << 32) | id;
  unsigned long long old;
  bool first;
  int slot = -1;
  if (cl.nslot > 0) {
    unsigned h = ((unsigned)node * 2654435761u) >> cl.shift;
    for (int pr = 0; pr < kCellProbes; pr++) {
      const int o = atomicCAS(&cl.lnode[h], -1, node);
      if (o == -1 || o == node) { slot = (int)h; first = (o == -1); break; }
      h = (h + 1) & (unsigned)(cl.nslot - 1);
    }
  }
  if (slot >= 0) {
    old = atomicMax(&cl.lkey[slot], key);
  } else {
    old = atomicMax(reinterpret_cast<unsigned long long *>(cl.ub + (unsigned)(cl.o_nodekey + 8u * (unsigned)node)), key);
    first = (old == 0ull);
  }
  const int s = wave_alloc(&sh.n_new, first);
  if (first) *reinterpret_cast<int2 *>(cl.ub + (unsigned)(cl.o_touched + 8u * (unsigned)s)) = make_int2(node, slot);
  // old == 0: nothing stored in the cell yet (the slot's claimer may still be on its way; it will
  // then see this key as its `old`, so no tie goes unnoticed)
  return (old != 0ull && (unsigned)(old >> 32) == (unsigned)(key >> 32) && old != key) ? old : 0ull;
}

// TIMED adds per-phase wall clocks (jamd_pass1_result.phase_us, thread 0; development aid selected
// with JAMD_BEAM_TIMING=1 when the work area is created) -- they cost some 20 VGPRs, so the
// production instantiation carries none.
#ifndef JAMD_BEAM_CB
#define JAMD_BEAM_CB 4                  // tokens per thread carried together through the finalize step
#endif
#ifndef JAMD_BEAM_WPE
#define JAMD_BEAM_WPE 4                 // waves per SIMD the register allocation targets (4 = one workgroup per CU)
#endif
template <bool TIMED, bool SVLDS>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(JAMD_BEAM_WPE, JAMD_BEAM_WPE)))
beam_pass1_kernel(LexDev lx, Work wk, const float *__restrict__ scores, int S,
                  const int *__restrict__ utt_off, int smode) {
  __shared__ Shared sh;
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  const int u = blockIdx.x, tid = threadIdx.x;
  if (tid == 0 && wk.resident) __hip_atomic_fetch_add(wk.resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // this workgroup holds its CU now
  // smode 0: whole utterances.  smode 1 / 2: streaming -- this launch advances every utterance
  // by the rows utt_off[u]..utt_off[u+1]) of `scores`; 2 = also run get_back_trellis_end() and
  // the traceback.  The state between launches lives in wk.stream[u] and in the slice at o_sv.
  const int t_begin = utt_off[u], nrows = utt_off[u + 1] - t_begin;
  StreamState *ss = smode ? wk.stream + u : nullptr;
  const bool resume = smode && ss->started;
  const int base = resume ? ss->frames_done : 0;            // absolute index of this launch's first row
  const int T = base + nrows;                                // frames seen so far
  const bool finish = smode != 1;                            // run the end phase after the last row
  unsigned char *const ub = wk.slices + (size_t)u * wk.utt_stride;     // this utterance's slice
#define SLICE(T, off, i) (*reinterpret_cast<T *>(ub + (unsigned)((off) + (unsigned)sizeof(T) * (unsigned)(i))))
#define NODEKEY(i) SLICE(unsigned long long, wk.o_nodekey, i)
#define CUR(i) SLICE(Tok, wk.o_cur, i)
#define CURKEY(i) SLICE(unsigned, wk.o_cur_key, i)
#define TOUCHED(i) SLICE(int2, wk.o_touched, i)
#define ARCQ(i) SLICE(int2, wk.o_arcq, i)            /* extra arcs of this frame's survivors: (survivor, arc) */
#define ATOM(i) SLICE(jamd_trellis_atom, wk.o_atoms, i)
  jamd_pass1_result *res = wk.res + u;
  // survivor state of the previous frame (tokens, the atom each word end emitted, the
  // frame's word-end list, node -> survivor hash)
  SvImage<SVLDS> svi;                       // SVLDS == (wk.use_lds != 0): chosen at launch
  svi.bind(dyn_lds, ub + wk.o_sv, wk.beam, wk.hsize);
  const auto sv_atom = svi.atom;
  const auto welist = svi.we;
  const auto hkey = svi.hkey;
  const auto hval = svi.hval;
  const int hmask = wk.hsize - 1;
  // the frame's Viterbi cells: LDS table behind the survivor image (16-byte aligned), see Cells
  Cells cl;
  cl.ub = ub; cl.o_nodekey = wk.o_nodekey; cl.o_touched = wk.o_touched; cl.nslot = SVLDS ? wk.cell_slots : 0;
  cl.lkey = (unsigned long long *)(dyn_lds + wk.cell_off);
  cl.lnode = (int *)(dyn_lds + wk.node_off);
  unsigned *hist = (unsigned *)(dyn_lds + wk.cell_off);    // step D only: the cells are all empty then
  float *rowc = (float *)(dyn_lds + wk.row_off);           // this frame's score row when wk.row_cache
  cl.shift = cl.nslot > 0 ? 32 - (31 - __clz(cl.nslot)) : 0;
  for (int i = tid; i < cl.nslot; i += NT) { cl.lkey[i] = 0ull; cl.lnode[i] = -1; }
  const float lmw = lx.lm_weight, pen = lx.lm_penalty;
  const bool dfa = lx.lm_type != JAMD_LM_NGRAM;          // grammar or word list: initial-token frame, no factoring
  const bool wordmode = lx.lm_type == JAMD_LM_WORD;      // isolated words: no cross-word transition at all
  unsigned long long *memo = reinterpret_cast<unsigned long long *>(ub + wk.o_lmcache);

  if (resume) {
    if (!ss->active) return;                                 // died / overflowed / finished earlier
    if (SVLDS) {                                             // survivor image back into LDS
      const u32x4 *src = (const u32x4 *)(ub + wk.o_sv);
      lds_v4 *dst = (lds_v4 *)dyn_lds;
      for (int i = tid; i < wk.sv_bytes / 16; i += NT) dst[i] = src[i];
    }
    if (tid == 0) {
      sh.n_atom = ss->n_atom; sh.ties = ss->ties; sh.ties_we = ss->ties_we; sh.ties_cut = ss->ties_cut;
      sh.n_surv = ss->n_surv;
    }
    __syncthreads();
  } else {
    if (tid == 0) {
      sh.n_atom = 0; sh.ties = 0; sh.ties_we = 0; sh.ties_cut = 0; sh.n_surv = 0;
      res->status = JAMD_PASS1_OK; res->natom = 0; res->wnum = 0; res->score = JAMD_LOG_ZERO;
      res->died_at = -1; res->ties = 0; res->frames = T; res->max_tokens = 0;
      for (int i = 0; i < 8; i++) res->phase_us[i] = 0;
    }
    for (int i = tid; i < wk.hsize; i += NT) hkey[i] = -1;
    for (int i = tid; i < wk.nscword; i += NT) memo[i] = 0xffffffff00000000ull;   // context -1: never matches
    __syncthreads();
    if (nrows <= 0) {                                        // nothing to start from yet
      if (tid == 0) { if (smode != 1) res->status = JAMD_PASS1_FAIL; if (ss) { ss->started = 0; ss->active = 1; } }
      return;
    }
    // ---- get_back_trellis_init(): the silB head token (init_nodescore, beam.c:1622-1665).
    // With a grammar the initial tokens (one per word that may start a sentence, :1669-1757)
    // enter through the finalize and rank-pruning steps of a pseudo frame 0 below.
    if (tid == 0 && !dfa) {
      const int node = lx.word_head(lx.head_silwid);
      const int4 nr = lx.node_b(node);                 // {stend, scid, out_id, out_kind}
      Tok nw;
      float ls = (nr.y != 0) ? max_successor_prob(lx, -1, nr.y) : 0.0f;
      ls = ls * lmw + pen;
      nw.node = node; nw.last_tre = -1; nw.last_cword = -1; nw.last_wid = -1; nw.last_lscore = ls;
      nw.score = node_outprob(lx, scores + (size_t)t_begin * S, nr.w, nr.z, -1) + ls;
      nw.pad0 = nw.pad1 = 0;
      svi.store(0, nw);
      hash_put(hkey, hval, hmask, node, 0);
      sh.n_surv = 1;
    }
  }
  float thr = resume ? ss->thr : JAMD_LOG_ZERO;        // d->score_pruning_threshold (beam.c:1935)
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = wall_clock64(), tq = 0;   // phase clocks (100 MHz), thread 0 only
#define PHASE(i) do { if (TIMED && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[i] += n_ - tc; tc = n_; } } while (0)
  int max_tokens = resume ? ss->max_tokens : 1;
  bool stopped = false;
  __syncthreads();

  // frames base+1 .. T-1 of this launch are propagated into (frame `base` itself when resuming);
  // t == T is the end phase and only runs when finishing
  for (int t = resume ? base : (dfa ? 0 : 1); t <= (finish ? T : T - 1); t++) {
    // tl/tn swap (beam.c:2697-2698): sv[] holds last frame's survivors
    const int n_surv = sh.n_surv;
    __syncthreads();
    if (tid == 0) { sh.n_new = 0; sh.n_we = 0; sh.n_arc = 0; sh.we_best = 0ull; sh.maxbits = ord(JAMD_LOG_ZERO); sh.minbits = 0xffffffffu; }
    __syncthreads();
    const bool last = (t == T);     // get_back_trellis_end(): word ends only, no pruning test
    if (wk.row_cache && !last) {    // readers: step C of this frame, two barriers from here
      const float *__restrict__ rg = scores + (size_t)(t_begin + t - base) * S;
      for (int i = tid; i < S; i += NT) rowc[i] = rg[i];
    }

    // one intra-word candidate of token tk: score, LM factoring update, push, tie accounting
    auto intra_candidate = [&](const Tok &tk, int next_node, float a) {
      const int node = tk.node;
      float tmpsum = tk.score + a;
      const int nscid = (next_node != node) ? lx.scid(next_node) : 0;
      const bool fac = nscid != 0;
      if (fac) {
        const float ng = max_successor_prob(lx, tk.last_cword, nscid, memo) * lmw + pen;
        tmpsum -= tk.last_lscore;
        tmpsum += ng;
      }
      const unsigned long long tie = push(sh, cl, next_node, tmpsum, (unsigned)node);
      if (tie != 0ull) {
        // two different sources reach next_node with exactly the same score.
        // Harmless when both carry the same history (same predecessor atom, context
        // word and LM score) -- merging tree branches produce that; anything else is
        // a genuine tie, resolved by the larger source id and counted.
        bool same = false;
        if (((unsigned)tie >> 31) == 0u) {
          const Tok o = svi.load(hash_get(hkey, hval, hmask, (int)(unsigned)tie));
          // the LM score is recomputed from last_cword on entering a factoring
          // node from another node (see step C); otherwise it is inherited
          const bool re_o = next_node != o.node && lx.scid(next_node) != 0;
          same = o.last_tre == tk.last_tre && o.last_cword == tk.last_cword &&
                 (fac == re_o) && (fac || o.last_lscore == tk.last_lscore);
        }
        if (!same) atomicAdd(&sh.ties, 1);
      }
    };
    // ---- A: intra-word transitions + word-end atoms (main loop, beam.c:2838-2900)
    for (int j = tid; j < n_surv; j += NT) {
      const Tok tk = svi.load(j);
      const int node = tk.node;
      const int4 na = lx.node_a(node);               // {self_a, next_a, ac_off, ac_end}
      const int sword = lx.node_b(node).x;           // stend
      if (!last) {
        if (tk.score <= JAMD_LOG_ZERO) continue;
        if (tk.score < thr) continue;
        // beam_intra_word() :2154-2180 -> beam_intra_word_core() :2004-2135.  The self loop and
        // the `next` arc are handled here; the extra arcs of branching nodes (up to tens per
        // node) go to a work queue so that no lane walks them alone.
        const int e0 = na.z, e1 = na.w;
        if (e1 > e0) {
          const int base = atomicAdd(&sh.n_arc, e1 - e0);
          for (int e = e0; e < e1; e++) ARCQ(base + e - e0) = make_int2(j, e);
        }
        for (int k = 0; k < 2; k++) {
          int next_node; float a;
          if (k == 0) { next_node = node; a = __int_as_float(na.x); if (a == JAMD_LOG_ZERO) continue; }
          else { next_node = node + 1; a = __int_as_float(na.y); if (a == JAMD_LOG_ZERO) continue; }
          intra_candidate(tk, next_node, a);
        }
      }
      if (sword >= 0) {
        // save_trellis() :2209-2247
        const int ai = wave_alloc(&sh.n_atom, true);
        if (ai < wk.atom_cap) {
          jamd_trellis_atom a;
          a.wid = sword; a.last_tre = tk.last_tre; a.backscore = tk.score; a.lscore = tk.last_lscore;
          a.begintime = (short)((tk.last_tre < 0 ? -1 : ATOM(tk.last_tre).endtime) + 1);
          a.endtime = (short)(t - 1);
          ATOM(ai) = a;
        }
        sv_atom[j] = ai;
        if (!last && !wordmode && sword != lx.tail_silwid) {   // beam_inter_word() :2296-2313
          welist[atomicAdd(&sh.n_we, 1)] = j;
          const float tmpprob = tk.score + lx.wordend_a(sword);
          if (!dfa && tmpprob > JAMD_LOG_ZERO) {
            const unsigned long long key = ((unsigned long long)ord(tmpprob) << 32) | (unsigned)sword;
            const unsigned long long old = atomicMax(&sh.we_best, key);
            if (old != 0ull && (unsigned)(old >> 32) == (unsigned)(key >> 32)) atomicAdd(&sh.ties_we, 1);
          }
        }
      }
    }
    __syncthreads();
    if (!last) {                                   // drain the extra-arc queue, one arc per thread
      const int n_arc = sh.n_arc;
      for (int q = tid; q < n_arc; q += NT) {
        const int2 it = ARCQ(q);
        intra_candidate(svi.load(it.x), lx.ac_to(it.y), lx.ac_a(it.y));
      }
      __syncthreads();
    }
    PHASE(0);
    if (last) break;

    // ---- B1: word ends -> isolated roots with the 2-gram (beam_inter_word() :2334-2516)
    if (dfa) {
      // grammar: every word end x every root whose category may follow (category-pair
      // constraint, beam_inter_word() :2404-2412), word insertion penalty + the ended word's
      // in-class score as the LM score (:2452-2461)
      const int n_we = sh.n_we, nroot = lx.startnum;
      const int total = n_we * nroot;
      for (int x = tid; x < total; x += NT) {
        const int w = x / nroot, r = x - w * nroot;
        const Tok tk = svi.load(welist[w]);
        const int sword = lx.node_b(tk.node).x;
        if (!lx.cat_pair(lx.wton(sword) * lx.ncat + lx.root_cat(r))) continue;
        const int last_word = lx.is_transparent(sword) ? tk.last_cword : sword;
        float tmpsum = tk.score;
        tmpsum += lx.wordend_a(sword);
        float ng = lx.penalty1;
        ng += (last_word >= 0) ? lx.cprob(last_word) : 0.0f;
        tmpsum += ng;
        if (push(sh, cl, lx.startnode(r), tmpsum, 0x80000000u | (unsigned)sword) != 0ull)
          atomicAdd(&sh.ties, 1);
      }
      if (t == 0)          // pseudo frame 0: the initial tokens (init_nodescore(), beam.c:1669-1757)
        for (int e = tid; e < lx.ninit; e += NT)
          push(sh, cl, lx.init_node(e), lx.init_lscore(e), 0xC0000000u | (unsigned)e);
    } else {
      const int n_we = sh.n_we, niso = lx.isolatenum;
      const int total = n_we * niso;
      for (int x = tid; x < total; x += NT) {
        const int w = x / niso, i = x - w * niso;
        const Tok tk = svi.load(welist[w]);
        const int sword = lx.node_b(tk.node).x;
        const bool tr = lx.is_transparent(sword) != 0;
        const int last_word = tr ? tk.last_cword : sword;
        const int2 ir = lx.iso_root(i);                    // {root node, successor word}
        // one entry of max_successor_prob_iw()'s array (factoring_sub.c:1119-1143)
        const float p = (last_word < 0) ? 0.0f
                        : lx.iwtab ? lx.iwtab[(size_t)lx.wton(last_word) * niso + i]
                        : bigram_prob(lx, lx.wton(last_word), lx.wton(ir.y)) + lx.cprob(ir.y);
        float tmpsum = tk.score;
        tmpsum += lx.wordend_a(sword);
        const float ng = p * lmw + pen;
        tmpsum += ng;
        if (tr && tk.last_cword >= 0 && lx.is_transparent(tk.last_cword)) tmpsum += lx.lm_penalty_trans;
        if (push(sh, cl, ir.x, tmpsum, 0x80000000u | (unsigned)sword) != 0ull)
          atomicAdd(&sh.ties, 1);
      }
    }
    // ---- B2: best word end -> shared roots with the 1-gram factoring value
    //          (beam_inter_word_factoring() :2549-2637)
    if (!dfa && sh.we_best != 0ull) {
      const unsigned long long kb = sh.we_best;
      const float best_score = unord((unsigned)(kb >> 32));
      const int sword = (int)(unsigned)kb;
      const Tok tk = svi.load(hash_get(hkey, hval, hmask, lx.word_end(sword)));
      const bool trans2 = lx.is_transparent(sword) && tk.last_cword >= 0 && lx.is_transparent(tk.last_cword);
      for (int r = tid; r < lx.nshared; r += NT) {
        const float2 sr = lx.shared_root(r);               // {root node (bits), fscore}
        const float ng = sr.y * lmw + pen;
        float tmpsum = best_score;
        tmpsum += ng;
        if (trans2) tmpsum += lx.lm_penalty_trans;
        if (tmpsum < thr) continue;                               // :2580
        if (push(sh, cl, __float_as_int(sr.x), tmpsum, 0xC0000000u) != 0ull) atomicAdd(&sh.ties, 1);
      }
    }
    __syncthreads();
    if (tid == 0) sh.n_arc = 0;                    // the queue now collects state-set reductions
    PHASE(1);

    // ---- C: finalize the touched nodes: winner's payload + acoustic score (:2944-2951)
    const int n_new = sh.n_new;
    if (n_new > max_tokens) max_tokens = n_new;
    {
      const RowRef row{scores + (size_t)(t_begin + t - base) * S, rowc, wk.row_cache != 0};
      unsigned mymax = ord(JAMD_LOG_ZERO), mymin = 0xffffffffu;
      // CB tokens per thread are carried through the steps together: every step's loads (node
      // record, LM memo, context table, score row) are issued for all of them before any is used, so
      // one thread has up to CB independent gathers in flight instead of one dependent chain per token.
      constexpr int CB = JAMD_BEAM_CB;
      for (int s0 = tid; s0 < n_new; s0 += CB * NT) {
        bool ok[CB]; int node[CB], slot[CB]; int4 nr[CB]; unsigned long long key[CB];
        int l_tre[CB], l_cword[CB], l_wid[CB], lmreq[CB], ent[CB];
        float l_ls[CB];
#pragma unroll
        for (int k = 0; k < CB; k++) {
          const int s = s0 + k * NT;
          ok[k] = s < n_new;
          const int2 t2 = ok[k] ? TOUCHED(s) : make_int2(0, -1);     // {node, LDS slot or -1}
          node[k] = t2.x; slot[k] = t2.y;
        }
#pragma unroll
        for (int k = 0; k < CB; k++) nr[k] = lx.node_b(node[k]);     // {stend, scid, out_id, out_kind}
#pragma unroll
        for (int k = 0; k < CB; k++) {
          key[k] = 0ull;
          if (ok[k]) {
            if (slot[k] >= 0) { key[k] = cl.lkey[slot[k]]; cl.lkey[slot[k]] = 0ull; cl.lnode[slot[k]] = -1; }   // back to empty
            else key[k] = atomicExch(&NODEKEY(node[k]), 0ull);
          }
        }
        // winner's payload.  lmreq != 0: the LM factoring value has to be looked up (next step)
#pragma unroll
        for (int k = 0; k < CB; k++) {
          const unsigned id = (unsigned)key[k];
          lmreq[k] = 0; l_tre[k] = -1; l_cword[k] = -1; l_wid[k] = -1; l_ls[k] = 0.0f;
          if (!ok[k]) continue;
          if ((id >> 31) == 0u) {                      // intra-word, id = source node
            const Tok tk = svi.load(hash_get(hkey, hval, hmask, (int)id));
            l_tre[k] = tk.last_tre; l_cword[k] = tk.last_cword; l_wid[k] = tk.last_wid;
            if (node[k] != tk.node && nr[k].y != 0) lmreq[k] = nr[k].y;   // beam_intra_word_core() :2069-2082
            else l_ls[k] = tk.last_lscore;
          } else if (dfa && (id >> 30) == 3u) {        // an initial token of the grammar
            l_ls[k] = lx.init_lscore(id & 0x3fffffffu);
          } else {
            const bool iso = (id >> 30) == 2u;
            const int sword = iso ? (int)(id & 0x3fffffffu) : (int)(unsigned)sh.we_best;
            const int j = hash_get(hkey, hval, hmask, lx.word_end(sword));
            const Tok tk = svi.load(j);
            const int last_word = lx.is_transparent(sword) ? tk.last_cword : sword;
            l_tre[k] = sv_atom[j]; l_cword[k] = last_word; l_wid[k] = sword;
            if (dfa) {                                       // beam_inter_word() :2452-2461
              float ng = lx.penalty1;
              ng += (last_word >= 0) ? lx.cprob(last_word) : 0.0f;
              l_ls[k] = ng;
            } else if (iso) {                                // beam_inter_word() :2430-2438
              const int wn = lx.scword(nr[k].y);
              const float p = (last_word < 0) ? 0.0f
                              : bigram_prob(lx, lx.wton(last_word), lx.wton(wn)) + lx.cprob(wn);
              l_ls[k] = p * lmw + pen;
            } else {                                         // beam_inter_word_factoring() :2572-2573
              l_ls[k] = lx.fscore(-nr[k].y) * lmw + pen;
            }
          }
        }
        // LM factoring value on entering a branch node: max_successor_prob(), its memo read issued
        // for all CB tokens first (factoring_sub.c:942-1008)
        {
          int ctx[CB]; unsigned long long mm[CB]; float fs[CB];
#pragma unroll
          for (int k = 0; k < CB; k++) {
            ctx[k] = -1; mm[k] = 0ull; fs[k] = 0.0f;
            if (lmreq[k] != 0 && l_cword[k] >= 0) {
              if (lmreq[k] < 0) fs[k] = lx.fscore(-lmreq[k]);
              else { ctx[k] = lx.wton(l_cword[k]); mm[k] = memo[lmreq[k]]; }
            }
          }
#pragma unroll
          for (int k = 0; k < CB; k++) {
            if (lmreq[k] == 0) continue;
            float p = 0.0f;                                   // lastword < 0: no LM context yet
            if (l_cword[k] >= 0) {
              if (lmreq[k] < 0) p = fs[k];
              else if ((int)(unsigned)(mm[k] >> 32) == ctx[k]) p = __uint_as_float((unsigned)mm[k]);
              else p = max_successor_prob(lx, l_cword[k], lmreq[k], memo);     // memo miss: 2-gram search, refill
            }
            l_ls[k] = p * lmw + pen;
          }
        }
        // outprob_style(), outprob_style.c:354-486: a plain state score is added here; a
        // state-set reduction (tens of gathers) is deferred to the cooperative drain below
        {
          int col[CB];
#pragma unroll
          for (int k = 0; k < CB; k++) {
            col[k] = lx.nlc;
            if (ok[k] && nr[k].w >= JAMD_AS_RSET && l_wid[k] >= 0) col[k] = lx.word_lc(l_wid[k]);
          }
#pragma unroll
          for (int k = 0; k < CB; k++) {
            if (nr[k].w == JAMD_AS_STATE) ent[k] = nr[k].z;
            else if (nr[k].w == JAMD_AS_LSET) ent[k] = ~nr[k].z;
            else ent[k] = ok[k] ? lx.lc_tab((size_t)nr[k].z * (lx.nlc + 1) + col[k]) : 0;
          }
        }
        float ac[CB];
#pragma unroll
        for (int k = 0; k < CB; k++) ac[k] = (ok[k] && ent[k] >= 0) ? row[ent[k]] : 0.0f;
#pragma unroll
        for (int k = 0; k < CB; k++) {
          if (!ok[k]) continue;
          const int s = s0 + k * NT;
          const float score = unord((unsigned)(key[k] >> 32));
          Tok nw;
          nw.node = node[k]; nw.pad0 = nw.pad1 = 0;
          nw.last_tre = l_tre[k]; nw.last_cword = l_cword[k]; nw.last_wid = l_wid[k]; nw.last_lscore = l_ls[k];
          if (ent[k] >= 0) {
            nw.score = score + ac[k];
            const unsigned b = ord(nw.score);
            CURKEY(s) = b;
            if (b > mymax) mymax = b;
            if (b < mymin) mymin = b;
          } else {
            nw.score = score;
            ARCQ(atomicAdd(&sh.n_arc, 1)) = make_int2(s, ~ent[k]);     // (token, state set); arcq is free again
          }
          CUR(s) = nw;
        }
      }
      __syncthreads();
      if (TIMED && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[4] += n_ - tc; tq = n_; }   // token loop
      // drain: four lanes per (token, set) item, each reduces every fourth member, then the
      // partial results are merged through shuffles (outprob_cd(), outprob.c:287-400)
      const int n_set = sh.n_arc;
      const int sub = tid & 3, lane = tid & 63;
      for (int q0 = 0; q0 < n_set; q0 += NT / 4) {
        const int q = q0 + (tid >> 2);
        const bool act = q < n_set;
        const int2 it = act ? ARCQ(q) : make_int2(0, 0);
        const int a = act ? lx.set_off(it.y) : 0, bnd = act ? lx.set_off(it.y + 1) : 0;
        float r;
        if (lx.cdset_method == JAMD_IWCD_NBEST && lx.cdmax_num <= 4) {
          float b0 = JAMD_LOG_ZERO, b1 = JAMD_LOG_ZERO, b2 = JAMD_LOG_ZERO, b3 = JAMD_LOG_ZERO;
          int n = 0;
          auto ins = [&](float p) {
            float t_;
            if (p > b0) { t_ = b0; b0 = p; p = t_; }
            if (p > b1) { t_ = b1; b1 = p; p = t_; }
            if (p > b2) { t_ = b2; b2 = p; p = t_; }
            if (p > b3) { b3 = p; }
          };
          for (int m = a + sub; m < bnd; m += 16) {          // four members per lane in flight
            int ix[4]; float pv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) ix[j] = (m + 4 * j < bnd) ? lx.set_states(m + 4 * j) : -1;
#pragma unroll
            for (int j = 0; j < 4; j++) pv[j] = (ix[j] >= 0) ? row[ix[j]] : JAMD_LOG_ZERO;
#pragma unroll
            for (int j = 0; j < 4; j++) if (pv[j] > JAMD_LOG_ZERO) { n++; ins(pv[j]); }
          }
          // merge the four partial top lists into the group's first lane (values <= LOG_ZERO are
          // padding and never displace anything)
#pragma unroll
          for (int src = 1; src < 4; src++) {
            const int from = (lane & ~3) + src;
            const float c0 = __shfl(b0, from, 64), c1 = __shfl(b1, from, 64), c2 = __shfl(b2, from, 64),
                        c3 = __shfl(b3, from, 64);
            const int cn = __shfl(n, from, 64);
            if (sub == 0) { ins(c0); ins(c1); ins(c2); ins(c3); n += cn; }
          }
          if (n > lx.cdmax_num) n = lx.cdmax_num;
          float sum = 0.0f;
          if (n > 0) sum += b0;
          if (n > 1) sum += b1;
          if (n > 2) sum += b2;
          if (n > 3) sum += b3;
          r = sum / (float)n;
        } else if (lx.cdset_method == JAMD_IWCD_MAX) {
          float m_ = JAMD_LOG_ZERO;
          for (int m = a + sub; m < bnd; m += 16) {
            int ix[4]; float pv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) ix[j] = (m + 4 * j < bnd) ? lx.set_states(m + 4 * j) : -1;
#pragma unroll
            for (int j = 0; j < 4; j++) pv[j] = (ix[j] >= 0) ? row[ix[j]] : JAMD_LOG_ZERO;
#pragma unroll
            for (int j = 0; j < 4; j++) if (m_ < pv[j]) m_ = pv[j];
          }
#pragma unroll
          for (int src = 1; src < 4; src++) { const float c = __shfl(m_, (lane & ~3) + src, 64); if (m_ < c) m_ = c; }
          r = m_;
        } else {
          // average (member-order float sum) and long N-best lists: one lane, reference order
          r = (act && sub == 0) ? cd_reduce(row, lx.set_states_ptr(), a, bnd, lx.cdset_method, lx.cdmax_num) : 0.0f;
        }
        if (act && sub == 0) {
          const float sc = CUR(it.x).score + r;
          CUR(it.x).score = sc;
          const unsigned b = ord(sc);
          CURKEY(it.x) = b;
          if (b > mymax) mymax = b;
          if (b < mymin) mymin = b;
        }
      }
      if (TIMED && tid == 0) { ph[5] += wall_clock64() - tq; ph[6] += sh.n_arc; }     // set drain; number of set reductions
      atomicMax(&sh.maxbits, mymax);
      atomicMin(&sh.minbits, mymin);
    }
    __syncthreads();
    PHASE(2);
    {
      const float mx = unord(sh.maxbits);                          // score_pruning_max :2948
      thr = (wk.width >= 0.0f) ? (mx - wk.width) : JAMD_LOG_ZERO;  // :2954-2960
      if (t == 0) thr = JAMD_LOG_ZERO;                             // get_back_trellis_init() sets no score threshold
    }
    if (n_new == 0) {                                              // :3012-3015
      if (tid == 0) { res->status = JAMD_PASS1_DIED; res->died_at = t; }
      stopped = true;
      __syncthreads();
      break;
    }
    if (sh.n_atom > wk.atom_cap) {
      if (tid == 0) res->status = JAMD_PASS1_OVERFLOW;
      stopped = true;
      __syncthreads();
      break;
    }

    // ---- D: rank pruning, sort_token_no_order() :1492 -> the top beam_width tokens become
    //         the next frame's survivors (copied into sv[], hashed by node)
    unsigned prefix = 0, need = 0, count_eq = 0;
    const bool prune = n_new > wk.beam;
    if (prune) {
      // radix select of the beam_width-th largest key.  All keys of a frame share their
      // high bits (scores lie within a few hundred log units), so only the bits below the
      // highest bit in which the frame's max and min differ are selected on, 11 at a time
      // (typically 2-3 passes).
      need = (unsigned)wk.beam;
      const unsigned diff = sh.maxbits ^ sh.minbits;
      int remaining = diff ? 32 - __clz(diff) : 0;       // number of varying low bits
      prefix = remaining < 32 ? (sh.maxbits >> remaining) : 0u;
      count_eq = (unsigned)n_new;
      while (remaining > 0) {
        const int w = remaining < 11 ? remaining : 11;
        const int shift = remaining - w;
        const unsigned dmask = (1u << w) - 1u;
        for (int i = tid; i < 2048; i += NT) hist[i] = 0;
        __syncthreads();
        for (int s = tid; s < n_new; s += NT) {
          const unsigned b = CURKEY(s);
          const unsigned hi = (shift + w < 32) ? (b >> (shift + w)) : 0u;
          if (hi == prefix) atomicAdd(&hist[(b >> shift) & dmask], 1u);
        }
        __syncthreads();
        {
          // suffix scan over the 2048 digits with the whole workgroup: thread i owns digits 2i, 2i+1;
          // `above` = tokens with a larger digit
          const unsigned h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
          const unsigned pair = h0 + h1;
          unsigned incl = pair;                          // inclusive suffix sum over the lanes >= this one
          const int ln = tid & 63;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_down(incl, off, 64);
            if (ln + off < 64) incl += o;
          }
          if (ln == 0) sh.wsum[tid >> 6] = incl;         // this wave's total
          __syncthreads();
          unsigned above = incl - pair;                  // larger digits inside the wave ...
          for (int wv = (tid >> 6) + 1; wv < NT / 64; wv++) above += sh.wsum[wv];   // ... and in the waves above
          // digit 2i+1 first (the larger one)
          if (above < need && need <= above + h1) { sh.sel_digit = 2u * tid + 1u; sh.sel_need = need - above; sh.sel_count = h1; }
          above += h1;
          if (above < need && need <= above + h0) { sh.sel_digit = 2u * tid; sh.sel_need = need - above; sh.sel_count = h0; }
        }
        __syncthreads();
        prefix = (prefix << w) | sh.sel_digit;
        need = sh.sel_need;
        count_eq = sh.sel_count;
        remaining -= w;
      }
      // prefix = score bits of the beam_width-th token; keep everything above it and
      // `need` of the count_eq tokens equal to it
      for (int i = tid; i < 2048; i += NT) hist[i] = 0;   // the histogram sat on the first cells: empty them again
    }
    for (int i = tid; i < wk.hsize; i += NT) hkey[i] = -1;
    const bool cut_tie = prune && count_eq > need;
    if (tid == 0) { sh.n_surv = 0; sh.eq_n = 0; if (cut_tie) sh.ties_cut += 1; }
    __syncthreads();
    if (cut_tie) {
      // several tokens share the cut score: collect their nodes (a handful), then keep
      // those on the smallest nodes (canonical; the reference keeps whichever its heap
      // order left inside)
      for (int s = tid; s < n_new; s += NT)
        if (CURKEY(s) == prefix) {
          const int q = atomicAdd(&sh.eq_n, 1);
          if (q < 128) sh.eq_node[q] = CUR(s).node;
        }
      __syncthreads();
    }
    for (int s = tid; s < n_new; s += NT) {
      bool keep = true;
      if (prune) {
        const unsigned b = CURKEY(s);
        keep = b > prefix;
        if (b == prefix) {
          if (!cut_tie) keep = true;
          else {
            const int mynode = CUR(s).node;
            unsigned rank = 0;
            if (sh.eq_n <= 128) {
              for (int q = 0; q < sh.eq_n; q++) rank += (sh.eq_node[q] < mynode) ? 1u : 0u;
            } else {
              for (int q = 0; q < n_new; q++) rank += (CURKEY(q) == prefix && CUR(q).node < mynode) ? 1u : 0u;
            }
            keep = rank < need;
          }
        }
      }
      if (keep) {
        const Tok me = CUR(s);
        const int j = wave_alloc(&sh.n_surv, true);
        svi.store(j, me);
        hash_put(hkey, hval, hmask, me.node, j);
      }
    }
    __syncthreads();
    PHASE(3);
  }
  __syncthreads();

  if (smode == 1) {            // not finished: park the state for the next launch
    if (SVLDS && !stopped) {
      u32x4 *dst = (u32x4 *)(ub + wk.o_sv);
      const lds_v4 *src = (const lds_v4 *)dyn_lds;
      for (int i = tid; i < wk.sv_bytes / 16; i += NT) dst[i] = src[i];
    }
    if (tid == 0) {
      ss->started = 1; ss->active = stopped ? 0 : 1; ss->frames_done = T; ss->n_surv = sh.n_surv; ss->thr = thr;
      ss->n_atom = sh.n_atom; ss->ties = sh.ties; ss->ties_we = sh.ties_we; ss->ties_cut = sh.ties_cut;
      ss->max_tokens = max_tokens;
      res->natom = min(sh.n_atom, wk.atom_cap); res->frames = T; res->max_tokens = max_tokens;
      res->ties = sh.ties + sh.ties_we + sh.ties_cut;
      if (TIMED) for (int i = 0; i < 8; i++) res->phase_us[i] += (int)(ph[i] / 100ull);
    }
    return;
  }
  if (ss && tid == 0) { ss->active = 0; ss->started = 1; ss->frames_done = T; }

  // ---- find_1pass_result() :399-431 + trace_backptr() :294-340
  const int natom = min(sh.n_atom, wk.atom_cap);
  if (tid == 0) sh.best_atom = -1;
  __syncthreads();
  if (res->status == JAMD_PASS1_OK && dfa) {
    // grammar (:433-455): the best word on the latest frame where a word survived; equal scores
    // go to the smaller word id (rw[t] is sorted by word id and the test is a strict <)
    if (tid == 0) { sh.n_arc = -1; sh.we_best = 0ull; }
    __syncthreads();
    int lt = -1;
    for (int i = tid; i < natom; i += NT)
      if (ATOM(i).backscore > JAMD_LOG_ZERO && ATOM(i).endtime > lt) lt = ATOM(i).endtime;
    if (lt >= 0) atomicMax(&sh.n_arc, lt);
    __syncthreads();
    lt = sh.n_arc;
    for (int i = tid; i < natom; i += NT)
      if (ATOM(i).endtime == lt && ATOM(i).backscore > JAMD_LOG_ZERO)
        atomicMax(&sh.we_best, ((unsigned long long)ord(ATOM(i).backscore) << 32) | (0xffffffffu - (unsigned)ATOM(i).wid));
    __syncthreads();
    const unsigned long long kb = sh.we_best;
    for (int i = tid; i < natom; i += NT)      // (frame, word) names one atom: a word has one end node
      if (kb != 0ull && ATOM(i).endtime == lt && (unsigned)ATOM(i).wid == 0xffffffffu - (unsigned)kb &&
          ord(ATOM(i).backscore) == (unsigned)(kb >> 32)) sh.best_atom = i;
  } else if (res->status == JAMD_PASS1_OK) {
    int best = -1;
    for (int i = tid; i < natom; i += NT)
      if (ATOM(i).wid == lx.tail_silwid && ATOM(i).backscore > JAMD_LOG_ZERO) best = i;  // ascending i
    if (best >= 0) atomicMax(&sh.best_atom, best);
  }
  __syncthreads();
  if (tid == 0) {
    res->natom = natom; res->ties = sh.ties + sh.ties_we + sh.ties_cut; res->max_tokens = max_tokens;
    res->ties_node = sh.ties; res->ties_wordend = sh.ties_we; res->ties_cut = sh.ties_cut;
    if (TIMED) for (int i = 0; i < 8; i++) res->phase_us[i] += (int)(ph[i] / 100ull);
    res->frames = T;
    if (sh.n_atom > wk.atom_cap) res->status = JAMD_PASS1_OVERFLOW;
    if (res->status == JAMD_PASS1_OK) {
      const int best = sh.best_atom;
      if (best < 0) res->status = JAMD_PASS1_FAIL;
      else {
        int n = 0, a = best;
        int rev[MAXSEQ];
        rev[n++] = ATOM(a).wid;
        while (ATOM(a).begintime > 0 && n < MAXSEQ) { a = ATOM(a).last_tre; rev[n++] = ATOM(a).wid; }
        for (int k = 0; k < n; k++) res->wseq[k] = rev[n - 1 - k];
        res->wnum = n; res->score = ATOM(best).backscore;
      }
    }
  }
}


#undef SLICE
#undef NODEKEY
#undef CUR
#undef CURKEY
#undef TOUCHED
#undef ARCQ
#undef ATOM

// ---------------------------------------------------------------------------------------
// STRICT-ORDER first pass (verification mode, jamd_beam_set_strict_order()).
//
// The reference resolves exact score ties by its visiting order, which is the output of
// a partial heap sort over token indices (beam.c:1342-1516) applied frame after frame; no
// parallel schedule can reproduce that.  This kernel therefore runs the reference's
// SEQUENTIAL algorithm -- same token creation order, same heap permutation, same
// first-writer-wins propagation -- with ONE LANE PER UTTERANCE (parallel only across the
// utterances of a batch).  It is two to three orders of magnitude slower per utterance
// than beam_pass1_kernel and exists so that the word trellis can be checked bit for bit
// against the reference in every case, ties included.  Same inputs, same result records.
struct STok { int last_tre, last_cword; float last_lscore, score; int node; int to_state; };   // to_state: forward-DFA state (TOKEN2.to_state), 0 without one

struct StrictWork {
  STok *tl[2];     // [utt][cap]   tlist[2]
  int *ti[2];      // [utt][cap]   tindex[2]
  int *token;      // [utt][nnode] node -> token id of the current list (-1 none)
  int cap;
};

struct SBeam {
  const LexDev *lx; const float *sc; int S;
  STok *tl[2]; int *ti[2]; int tnum[2]; int *token; int cap;
  int tn, tlx, n_start, n_end;
  float thr, we_best_score; int we_best_node, we_best_tre, we_best_cword;
  jamd_trellis_atom *atoms; int natom, atom_cap; bool overflow;
};

__device__ int s_create_token(SBeam &b) {                       // create_token() beam.c:1148
  const int id = b.tnum[b.tn];
  if (id + 1 >= b.cap) { b.overflow = true; return id > 0 ? id - 1 : 0; }
  b.tnum[b.tn]++;
  b.ti[b.tn][id] = id;
  return id;
}

// sort_token_upward / _downward (beam.c:1342 / :1414): 1-based heap over tindex
__device__ void s_sort(SBeam &b, int neednum, int totalnum, bool upward) {
  STok *tl = b.tl[b.tn]; int *ti = b.ti[b.tn];
#define SD_(A) ti[(A) - 1]
#define SV_(A) (tl[ti[(A) - 1]].score)
#define BEFORE_(x, y) (upward ? ((x) < (y)) : ((x) > (y)))
#define STOP_(x, y) (upward ? ((x) >= (y)) : ((x) <= (y)))
  int n, root, child, parent, s;
  for (root = totalnum / 2; root >= 1; root--) {
    s = SD_(root); parent = root;
    while ((child = parent * 2) <= totalnum) {
      if (child < totalnum && BEFORE_(SV_(child), SV_(child + 1))) child++;
      if (STOP_(tl[s].score, SV_(child))) break;
      SD_(parent) = SD_(child); parent = child;
    }
    SD_(parent) = s;
  }
  n = totalnum;
  while (n > totalnum - neednum) {
    s = SD_(n); SD_(n) = SD_(1); n--; parent = 1;
    while ((child = parent * 2) <= n) {
      if (child < n && BEFORE_(SV_(child), SV_(child + 1))) child++;
      if (STOP_(tl[s].score, SV_(child))) break;
      SD_(parent) = SD_(child); parent = child;
    }
    SD_(parent) = s;
  }
#undef SD_
#undef SV_
#undef BEFORE_
#undef STOP_
}
__device__ void s_sort_no_order(SBeam &b, int neednum) {         // sort_token_no_order() :1492
  const int totalnum = b.tnum[b.tn], restnum = totalnum - neednum;
  if (neednum >= totalnum) { b.n_start = 0; b.n_end = totalnum - 1; }
  else if (neednum < restnum) { s_sort(b, neednum, totalnum, true); b.n_start = totalnum - neednum; b.n_end = totalnum - 1; }
  else { s_sort(b, restnum, totalnum, false); b.n_start = 0; b.n_end = neednum - 1; }
}

__device__ void s_propagate(SBeam &b, int next_node, float next_score, int last_tre, int last_cword,
                            float last_lscore, int to_state = 0) {   // propagate_token() :1945
  if (next_score <= JAMD_LOG_ZERO) return;
  int id = b.token[next_node];
  if (id >= 0) {
    STok &tk = b.tl[b.tn][id];
    if (tk.score < next_score) { tk.last_tre = last_tre; tk.last_cword = last_cword; tk.last_lscore = last_lscore; tk.score = next_score; tk.to_state = to_state; }
  } else {
    id = s_create_token(b);
    STok &tk = b.tl[b.tn][id];
    tk.last_tre = last_tre; tk.last_cword = last_cword; tk.last_lscore = last_lscore; tk.score = next_score;
    tk.node = next_node; tk.to_state = to_state; b.token[next_node] = id;
  }
}

__device__ void s_intra_core(SBeam &b, const STok &tk, int next_node, float next_a) {   // :2004
  const LexDev &lx = *b.lx;
  float tmpsum = tk.score + next_a, ng = JAMD_LOG_ZERO;
  const int nscid = (next_node != tk.node) ? lx.scid(next_node) : 0;
  if (nscid != 0) {
    ng = max_successor_prob(lx, tk.last_cword, nscid) * lx.lm_weight + lx.lm_penalty;
    tmpsum -= tk.last_lscore;
    tmpsum += ng;
  }
  if (ng == JAMD_LOG_ZERO) ng = tk.last_lscore;
  s_propagate(b, next_node, tmpsum, tk.last_tre, tk.last_cword, ng, tk.to_state);     // :2120
}

__device__ int s_save_trellis(SBeam &b, const STok &tk, int sword, int t) {            // :2209
  if (b.natom >= b.atom_cap) { b.overflow = true; return b.natom - 1; }
  jamd_trellis_atom a;
  a.wid = sword; a.backscore = tk.score; a.last_tre = tk.last_tre; a.lscore = tk.last_lscore;
  a.begintime = (short)((tk.last_tre < 0 ? -1 : b.atoms[tk.last_tre].endtime) + 1);
  a.endtime = (short)(t - 1);
  b.atoms[b.natom] = a;
  return b.natom++;
}

__global__ void __launch_bounds__(64)
beam_strict_kernel(LexDev lx, Work wk, StrictWork sw, const float *__restrict__ scores, int S,
                   const int *__restrict__ utt_off, int nutt) {
  const int u = blockIdx.x * 64 + threadIdx.x;
  if (u >= nutt) return;
  const int t_begin = utt_off[u], T = utt_off[u + 1] - t_begin;
  jamd_pass1_result *res = wk.res + u;
  SBeam b;
  b.lx = &lx; b.sc = scores + (size_t)t_begin * S; b.S = S;
  for (int i = 0; i < 2; i++) { b.tl[i] = sw.tl[i] + (size_t)u * sw.cap; b.ti[i] = sw.ti[i] + (size_t)u * sw.cap; b.tnum[i] = 0; }
  b.token = sw.token + (size_t)u * wk.nnode; b.cap = sw.cap;
  b.atoms = reinterpret_cast<jamd_trellis_atom *>(wk.slices + (size_t)u * wk.utt_stride + wk.o_atoms); b.natom = 0; b.atom_cap = wk.atom_cap; b.overflow = false;
  res->status = JAMD_PASS1_OK; res->natom = 0; res->wnum = 0; res->score = JAMD_LOG_ZERO; res->died_at = -1;
  res->ties = res->ties_node = res->ties_wordend = res->ties_cut = 0; res->frames = T; res->max_tokens = 0;
  for (int i = 0; i < 8; i++) res->phase_us[i] = 0;
  if (T <= 0) { res->status = JAMD_PASS1_FAIL; return; }
  for (int i = 0; i < wk.nnode; i++) b.token[i] = -1;               // init_nodescore() :1587-1590
  const float lmw = lx.lm_weight, pen = lx.lm_penalty;
  int status = JAMD_PASS1_OK, died_at = -1, max_tokens = 1;

  b.tn = 0; b.tlx = 1;
  const bool dfa = lx.lm_type != JAMD_LM_NGRAM;
  const bool wordmode = lx.lm_type == JAMD_LM_WORD;
  if (dfa) {                                                         // init_nodescore() :1669-1757, :1762-1788
    for (int e = 0; e < lx.ninit; e++) {
      const int id = s_create_token(b);
      STok &nw = b.tl[b.tn][id];
      const int node = lx.init_node(e);
      const int4 nr = lx.node_b(node);
      nw.last_lscore = lx.init_lscore(e); nw.last_tre = -1; nw.last_cword = -1;
      nw.score = node_outprob(lx, b.sc, nr.w, nr.z, -1) + nw.last_lscore;
      nw.node = node; nw.to_state = lx.nfwd ? lx.init_to_state(e) : 0; b.token[node] = id;       // :1739-1747
    }
  } else {                                                           // init_nodescore() :1622-1665
    const int id = s_create_token(b);
    STok &nw = b.tl[b.tn][id];
    const int node = lx.word_head(lx.head_silwid);
    const int4 nr = lx.node_b(node);
    float ls = (nr.y != 0) ? max_successor_prob(lx, -1, nr.y) : 0.0f;
    ls = ls * lmw + pen;
    nw.last_lscore = ls; nw.last_tre = -1; nw.last_cword = -1;
    nw.score = node_outprob(lx, b.sc, nr.w, nr.z, -1) + ls;
    nw.node = node; nw.to_state = 0; b.token[node] = id;
  }
  s_sort_no_order(b, wk.beam);
  b.thr = JAMD_LOG_ZERO;

  for (int t = 1; t < T; t++) {                                      // get_back_trellis_proceed() :2663
    b.tlx = b.tn; b.tn = b.tn ? 0 : 1;
    const int tl = b.tlx, tn = b.tn;
    b.we_best_score = JAMD_LOG_ZERO;
    for (int j = 0; j < b.tnum[tl]; j++) b.token[b.tl[tl][j].node] = -1;        // clear_tokens() :1122
    for (int j = b.n_start; j <= b.n_end; j++) {
      const STok tk = b.tl[tl][b.ti[tl][j]];
      if (tk.score <= JAMD_LOG_ZERO) continue;
      if (tk.score < b.thr) continue;
      const int node = tk.node;
      const int4 na = lx.node_a(node);
      const float a_self = __int_as_float(na.x), a_next = __int_as_float(na.y);
      if (a_self != JAMD_LOG_ZERO) s_intra_core(b, tk, node, a_self);           // beam_intra_word() :2154
      if (a_next != JAMD_LOG_ZERO) s_intra_core(b, tk, node + 1, a_next);
      for (int e = na.z; e < na.w; e++) s_intra_core(b, tk, lx.ac_to(e), lx.ac_a(e));
      const int sword = lx.node_b(node).x;
      if (sword >= 0) {
        const int tre = s_save_trellis(b, tk, sword, t);
        if (wordmode) {                                                         // :2875: isolated words stop here
        } else if (dfa) {                                                       // beam_inter_word(), grammar branch
          const int last_word = lx.is_transparent(sword) ? tk.last_cword : sword;
          for (int stid = lx.startnum - 1; stid >= 0; stid--) {
            if (!lx.cat_pair(lx.wton(sword) * lx.ncat + lx.root_cat(stid))) continue;      // :2404-2412
            int next_state = 0;
            if (lx.nfwd) { next_state = fwd_next(lx, tk.to_state, lx.root_cat(stid)); if (next_state < 0) continue; }   // :2412-2422
            float tmpsum = tk.score;
            tmpsum += lx.wordend_a(sword);
            float ng = lx.penalty1;                                             // :2452-2461
            ng += (last_word >= 0) ? lx.cprob(last_word) : 0.0f;
            tmpsum += ng;
            s_propagate(b, lx.startnode(stid), tmpsum, tre, last_word, ng, next_state);
          }
        } else if (sword != lx.tail_silwid) {                                   // beam_inter_word() :2271
          const bool tr = lx.is_transparent(sword) != 0;
          const int last_word = tr ? tk.last_cword : sword;
          float tmpprob = tk.score + lx.wordend_a(sword);
          if (b.we_best_score < tmpprob) {
            b.we_best_score = tmpprob; b.we_best_node = node; b.we_best_tre = tre; b.we_best_cword = tk.last_cword;
          }
          for (int stid = lx.startnum - 1; stid >= 0; stid--) {
            if (lx.start2isolate(stid) == -1) continue;
            const int next_node = lx.startnode(stid);
            const int wn = lx.scword(lx.scid(next_node));
            const float p = (last_word < 0) ? 0.0f
                            : bigram_prob(lx, lx.wton(last_word), lx.wton(wn)) + lx.cprob(wn);
            float tmpsum = tk.score;
            tmpsum += lx.wordend_a(sword);
            const float ng = p * lmw + pen;
            tmpsum += ng;
            if (tr && tk.last_cword >= 0 && lx.is_transparent(tk.last_cword)) tmpsum += lx.lm_penalty_trans;
            s_propagate(b, next_node, tmpsum, tre, last_word, ng);
          }
        }
      }
    }
    if (!dfa && b.we_best_score > JAMD_LOG_ZERO) {                               // beam_inter_word_factoring() :2549
      const int sword = lx.node_b(b.we_best_node).x;
      const int last_word = lx.is_transparent(sword) ? b.we_best_cword : sword;
      for (int stid = lx.startnum - 1; stid >= 0; stid--) {
        if (lx.start2isolate(stid) != -1) continue;
        const int next_node = lx.startnode(stid);
        const float ng = lx.fscore(-lx.scid(next_node)) * lmw + pen;
        float tmpsum = b.we_best_score;
        tmpsum += ng;
        if (lx.is_transparent(sword) && b.we_best_cword >= 0 && lx.is_transparent(b.we_best_cword)) tmpsum += lx.lm_penalty_trans;
        if (tmpsum < b.thr) continue;
        s_propagate(b, next_node, tmpsum, b.we_best_tre, last_word, ng);
      }
    }
    float pmax = JAMD_LOG_ZERO;
    const float *row = b.sc + (size_t)t * S;
    for (int j = 0; j < b.tnum[tn]; j++) {                                       // :2944-2951
      STok &tk = b.tl[tn][b.ti[tn][j]];
      const int4 nr = lx.node_b(tk.node);
      const int lw = tk.last_tre < 0 ? -1 : b.atoms[tk.last_tre].wid;
      tk.score += node_outprob(lx, row, nr.w, nr.z, lw);
      if (pmax < tk.score) pmax = tk.score;
    }
    b.thr = (wk.width >= 0.0f) ? (pmax - wk.width) : JAMD_LOG_ZERO;
    if (b.tnum[tn] > max_tokens) max_tokens = b.tnum[tn];
    b.tnum[tl] = 0;
    s_sort_no_order(b, wk.beam);
    if (b.tnum[tn] == 0) { status = JAMD_PASS1_DIED; died_at = t; break; }
    if (b.overflow) break;
  }
  if (status == JAMD_PASS1_OK && !b.overflow) {                                  // get_back_trellis_end() :3076
    b.tlx = b.tn; b.tn = b.tn ? 0 : 1;
    for (int j = b.n_start; j <= b.n_end; j++) {
      const STok tk = b.tl[b.tlx][b.ti[b.tlx][j]];
      const int sword = lx.node_b(tk.node).x;
      if (sword >= 0) s_save_trellis(b, tk, sword, T);
    }
    int best = -1;                                                               // find_1pass_result() :399
    if (dfa) {                                                                   // :433-455
      int lt = -1;
      for (int i = b.natom - 1; i >= 0 && lt < 0; i--) if (b.atoms[i].backscore > JAMD_LOG_ZERO) lt = b.atoms[i].endtime;
      for (int i = 0; i < b.natom; i++) {        // atoms are emitted in time order
        const jamd_trellis_atom &a = b.atoms[i];
        if (a.endtime != lt || !(a.backscore > JAMD_LOG_ZERO)) continue;
        if (best < 0 || b.atoms[best].backscore < a.backscore ||
            (b.atoms[best].backscore == a.backscore && a.wid < b.atoms[best].wid)) best = i;
      }
    } else {
      // atoms are emitted in time order: the first hit from the back is the tail word ending latest
      for (int i = b.natom - 1; i >= 0; i--)
        if (b.atoms[i].wid == lx.tail_silwid && b.atoms[i].backscore > JAMD_LOG_ZERO) { best = i; break; }
    }
    if (best < 0) status = JAMD_PASS1_FAIL;
    else {
      int n = 0, a = best;
      int rev[MAXSEQ];
      rev[n++] = b.atoms[a].wid;
      while (b.atoms[a].begintime > 0 && n < MAXSEQ) { a = b.atoms[a].last_tre; rev[n++] = b.atoms[a].wid; }
      for (int k = 0; k < n; k++) res->wseq[k] = rev[n - 1 - k];
      res->wnum = n; res->score = b.atoms[best].backscore;
    }
  }
  if (b.overflow) status = JAMD_PASS1_OVERFLOW;
  res->status = status; res->died_at = died_at; res->natom = b.natom; res->max_tokens = max_tokens;
}

// ---- multipath lexicons (hmminfo->multipath), strict order only -------------------------------------
// A multipath model (model-skip / state-skip transitions) builds a lexicon whose word-begin and
// word-end nodes have no output, and beam.c runs a different frame for it (:2747-2836): word-internal
// transitions of every survivor, THEN the beam over the new tokens, THEN trellis words and cross-word
// transitions from the word ends among those (the root has no output, so the token is passed on along
// the root's own arcs within the frame, :2467-2510), output probabilities only on emitting nodes
// (:2930-2943); frame 0 already goes through this (pass1.c:239) and one transition-only call ends the
// input (:3066-3073).  Kept as its own kernel beside beam_strict_kernel: same helpers, same records.
// Exact by construction like its parent; the CPU restatement of the same frame is pinned to the reference on
// multipath tasks (tests/test_beam_oracle.py), the kernel against both (tests/test_beam_gpu.py::test_multipath_*).
__device__ void s_enter_word_mp(SBeam &b, const LexDev &lx, int root, float tmpsum, int tre, int last_word, float ng, int to_state = 0) {
  const int4 na = lx.node_a(root);
  const float a_self = __int_as_float(na.x), a_next = __int_as_float(na.y);
  if (a_self != JAMD_LOG_ZERO) s_propagate(b, root, tmpsum + a_self, tre, last_word, ng, to_state);
  if (a_next != JAMD_LOG_ZERO) s_propagate(b, root + 1, tmpsum + a_next, tre, last_word, ng, to_state);
  for (int e = na.z; e < na.w; e++) s_propagate(b, lx.ac_to(e), tmpsum + lx.ac_a(e), tre, last_word, ng, to_state);
}

__global__ void __launch_bounds__(64)
beam_strict_mp_kernel(LexDev lx, Work wk, StrictWork sw, const float *__restrict__ scores, int S,
                      const int *__restrict__ utt_off, int nutt) {
  const int u = blockIdx.x * 64 + threadIdx.x;
  if (u >= nutt) return;
  const int t_begin = utt_off[u], T = utt_off[u + 1] - t_begin;
  jamd_pass1_result *res = wk.res + u;
  SBeam b;
  b.lx = &lx; b.sc = scores + (size_t)t_begin * S; b.S = S;
  for (int i = 0; i < 2; i++) { b.tl[i] = sw.tl[i] + (size_t)u * sw.cap; b.ti[i] = sw.ti[i] + (size_t)u * sw.cap; b.tnum[i] = 0; }
  b.token = sw.token + (size_t)u * wk.nnode; b.cap = sw.cap;
  b.atoms = reinterpret_cast<jamd_trellis_atom *>(wk.slices + (size_t)u * wk.utt_stride + wk.o_atoms); b.natom = 0; b.atom_cap = wk.atom_cap; b.overflow = false;
  res->status = JAMD_PASS1_OK; res->natom = 0; res->wnum = 0; res->score = JAMD_LOG_ZERO; res->died_at = -1;
  res->ties = res->ties_node = res->ties_wordend = res->ties_cut = 0; res->frames = T; res->max_tokens = 0;
  for (int i = 0; i < 8; i++) res->phase_us[i] = 0;
  if (T <= 0) { res->status = JAMD_PASS1_FAIL; return; }
  for (int i = 0; i < wk.nnode; i++) b.token[i] = -1;
  const float lmw = lx.lm_weight, pen = lx.lm_penalty;
  int status = JAMD_PASS1_OK, died_at = -1, max_tokens = 1;
  b.tn = 0; b.tlx = 1;
  const bool dfa = lx.lm_type != JAMD_LM_NGRAM;
  const bool wordmode = lx.lm_type == JAMD_LM_WORD;
  const int head_root = dfa ? -1 : lx.word_head(lx.head_silwid);
  if (dfa) {                                                         // init_nodescore(): score = LM score only (:1733)
    for (int e = 0; e < lx.ninit; e++) {
      const int id = s_create_token(b);
      STok &nw = b.tl[b.tn][id];
      const int node = lx.init_node(e);
      nw.last_lscore = lx.init_lscore(e); nw.last_tre = -1; nw.last_cword = -1;
      nw.score = nw.last_lscore;
      nw.node = node; nw.to_state = lx.nfwd ? lx.init_to_state(e) : 0; b.token[node] = id;
    }
  } else {                                                           // :1635-1663
    const int id = s_create_token(b);
    STok &nw = b.tl[b.tn][id];
    const int4 nr = lx.node_b(head_root);
    float ls = (nr.y != 0) ? max_successor_prob(lx, -1, nr.y) : 0.0f;
    ls = ls * lmw + pen;
    nw.last_lscore = ls; nw.last_tre = -1; nw.last_cword = -1;
    nw.score = ls;
    nw.node = head_root; b.token[head_root] = id;
  }
  s_sort_no_order(b, wk.beam);
  b.thr = JAMD_LOG_ZERO;

  for (int t = 0; t <= T; t++) {                                     // t == T: get_back_trellis_end()'s final call
    const bool final = t == T;
    b.tlx = b.tn; b.tn = b.tn ? 0 : 1;
    const int tl = b.tlx, tn = b.tn;
    b.we_best_score = JAMD_LOG_ZERO;
    for (int j = 0; j < b.tnum[tl]; j++) b.token[b.tl[tl][j].node] = -1;
    for (int j = b.n_start; j <= b.n_end; j++) {                     // :2752-2769
      const STok tk = b.tl[tl][b.ti[tl][j]];
      if (tk.score <= JAMD_LOG_ZERO) continue;
      if (tk.score < b.thr) continue;
      const int node = tk.node;
      const int4 na = lx.node_a(node);
      const float a_self = __int_as_float(na.x), a_next = __int_as_float(na.y);
      if (a_self != JAMD_LOG_ZERO) s_intra_core(b, tk, node, a_self);
      if (a_next != JAMD_LOG_ZERO) s_intra_core(b, tk, node + 1, a_next);
      for (int e = na.z; e < na.w; e++) s_intra_core(b, tk, lx.ac_to(e), lx.ac_a(e));
    }
    s_sort_no_order(b, wk.beam);                                     // :2774, over the new tokens
    for (int j = b.n_start; j <= b.n_end; j++) {                     // :2779-2825
      const STok tk = b.tl[tn][b.ti[tn][j]];
      if (tk.score < b.thr) continue;
      const int node = tk.node;
      const int sword = lx.node_b(node).x;
      if (sword < 0) continue;
      const int tre = s_save_trellis(b, tk, sword, t);
      if (final || wordmode) continue;
      if (dfa) {
        const int last_word = lx.is_transparent(sword) ? tk.last_cword : sword;
        for (int stid = lx.startnum - 1; stid >= 0; stid--) {
          if (!lx.cat_pair(lx.wton(sword) * lx.ncat + lx.root_cat(stid))) continue;
          int next_state = 0;
          if (lx.nfwd) { next_state = fwd_next(lx, tk.to_state, lx.root_cat(stid)); if (next_state < 0) continue; }   // :2412-2422
          float tmpsum = tk.score;
          float ng = lx.penalty1;
          ng += (last_word >= 0) ? lx.cprob(last_word) : 0.0f;
          tmpsum += ng;
          s_enter_word_mp(b, lx, lx.startnode(stid), tmpsum, tre, last_word, ng, next_state);
        }
      } else if (sword != lx.tail_silwid) {
        const bool tr = lx.is_transparent(sword) != 0;
        const int last_word = tr ? tk.last_cword : sword;
        if (b.we_best_score < tk.score) {                                       // no wordend_a in multipath (:2307)
          b.we_best_score = tk.score; b.we_best_node = node; b.we_best_tre = tre; b.we_best_cword = tk.last_cword;
        }
        for (int stid = lx.startnum - 1; stid >= 0; stid--) {
          const int next_node = lx.startnode(stid);
          if (next_node == head_root) continue;                                 // :2336-2341
          if (lx.start2isolate(stid) == -1) continue;
          const int wn = lx.scword(lx.scid(next_node));
          const float p = (last_word < 0) ? 0.0f
                          : bigram_prob(lx, lx.wton(last_word), lx.wton(wn)) + lx.cprob(wn);
          float tmpsum = tk.score;
          const float ng = p * lmw + pen;
          tmpsum += ng;
          if (tr && tk.last_cword >= 0 && lx.is_transparent(tk.last_cword)) tmpsum += lx.lm_penalty_trans;
          s_enter_word_mp(b, lx, next_node, tmpsum, tre, last_word, ng);
        }
      }
    }
    if (!dfa && b.we_best_score > JAMD_LOG_ZERO) {                               // beam_inter_word_factoring()
      const int sword = lx.node_b(b.we_best_node).x;
      const int last_word = lx.is_transparent(sword) ? b.we_best_cword : sword;
      for (int stid = lx.startnum - 1; stid >= 0; stid--) {
        const int next_node = lx.startnode(stid);
        if (next_node == head_root) continue;                                    // :2566-2571
        if (lx.start2isolate(stid) != -1) continue;
        const float ng = lx.fscore(-lx.scid(next_node)) * lmw + pen;
        float tmpsum = b.we_best_score;
        tmpsum += ng;
        if (lx.is_transparent(sword) && b.we_best_cword >= 0 && lx.is_transparent(b.we_best_cword)) tmpsum += lx.lm_penalty_trans;
        if (tmpsum < b.thr) continue;
        s_enter_word_mp(b, lx, next_node, tmpsum, b.we_best_tre, last_word, ng);
      }
    }
    float pmax = JAMD_LOG_ZERO;
    if (!final) {                                                                // :2930-2943
      const float *row = b.sc + (size_t)t * S;
      for (int j = 0; j < b.tnum[tn]; j++) {
        STok &tk = b.tl[tn][b.ti[tn][j]];
        const int4 nr = lx.node_b(tk.node);
        if (nr.w == JAMD_AS_NONE) continue;                                      // non-output node
        const int lw = tk.last_tre < 0 ? -1 : b.atoms[tk.last_tre].wid;
        tk.score += node_outprob(lx, row, nr.w, nr.z, lw);
        if (pmax < tk.score) pmax = tk.score;
      }
    }
    b.thr = (wk.width >= 0.0f) ? (pmax - wk.width) : JAMD_LOG_ZERO;
    if (b.tnum[tn] > max_tokens) max_tokens = b.tnum[tn];
    b.tnum[tl] = 0;
    s_sort_no_order(b, wk.beam);
    if (b.tnum[tn] == 0) { if (!final) { status = JAMD_PASS1_DIED; died_at = t; } break; }
    if (b.overflow) break;
  }
  if (status == JAMD_PASS1_OK && !b.overflow) {
    int best = -1;                                                               // find_1pass_result() :399
    if (dfa) {
      int lt = -1;
      for (int i = b.natom - 1; i >= 0 && lt < 0; i--) if (b.atoms[i].backscore > JAMD_LOG_ZERO) lt = b.atoms[i].endtime;
      for (int i = 0; i < b.natom; i++) {
        const jamd_trellis_atom &a = b.atoms[i];
        if (a.endtime != lt || !(a.backscore > JAMD_LOG_ZERO)) continue;
        if (best < 0 || b.atoms[best].backscore < a.backscore ||
            (b.atoms[best].backscore == a.backscore && a.wid < b.atoms[best].wid)) best = i;
      }
    } else {
      for (int i = b.natom - 1; i >= 0; i--)
        if (b.atoms[i].wid == lx.tail_silwid && b.atoms[i].backscore > JAMD_LOG_ZERO) { best = i; break; }
    }
    if (best < 0) status = JAMD_PASS1_FAIL;
    else {
      int n = 0, a = best;
      int rev[MAXSEQ];
      rev[n++] = b.atoms[a].wid;
      while (b.atoms[a].begintime > 0 && n < MAXSEQ) { a = b.atoms[a].last_tre; rev[n++] = b.atoms[a].wid; }
      for (int k = 0; k < n; k++) res->wseq[k] = rev[n - 1 - k];
      res->wnum = n; res->score = b.atoms[best].backscore;
    }
  }
  if (b.overflow) status = JAMD_PASS1_OVERFLOW;
  res->status = status; res->died_at = died_at; res->natom = b.natom; res->max_tokens = max_tokens;
}

// the cross-word LM table (LexDev::iwtab): one thread per (context, isolated root)
constexpr size_t kIwTabMaxBytes = (size_t)2 << 30;
__global__ void __launch_bounds__(256) iwtab_build_kernel(LexDev lx, float *tab, int nctx, int niso) {
  const size_t x = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= (size_t)nctx * niso) return;
  const int ctx = (int)(x / niso), i = (int)(x - (size_t)ctx * niso);
  const int w = lx.iso_root(i).y;
  tab[x] = bigram_prob(lx, ctx, lx.wton(w)) + lx.cprob(w);
}

template <typename T>
int upload(T **dst, const T *src, size_t n) {
  JAMD_HIP(hipMalloc((void **)dst, sizeof(T) * (n ? n : 1)));
  if (n) JAMD_HIP(hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
  return JAMD_OK;
}

}  // namespace

struct jamd_lexicon {
  jamd_engine *eng = nullptr;
  LexDev d{};
  int maxfan = 2, nscword = 0;
  bool multipath = false;          // JAMD_LM_MULTIPATH lexicon: its own frame (beam_exact_mp.h; strict order: beam_strict_mp_kernel)
  bool mp_parallel = false;        // ... and no root reaches a word-end node along its own arcs: the frame-parallel kernel can decode it
  // multipath: where a token entering a word goes (the root has no output: beam.c:2467-2510) -- one int4 {target node,
  // transition bits, root number * maxfan + transition number, root number / fscore bits} per transition a root really has,
  // in visiting order; byte offsets into the lexicon arena + entry counts (XWork carries them to the kernel)
  unsigned o_mp_iso = 0, o_mp_shared = 0, o_mp_start = 0;
  int n_mp_iso = 0, n_mp_shared = 0, n_mp_start = 0;
  // ... and the nodes those transitions lead to, numbered densely: int [nnode], -1 = never entered from a root.  Only such a
  // node can meet a token of the frame's second half, so the per-utterance "which token sits on this node" table of the
  // multipath frame (XWork::o_nodetok) has n_mp_tgt entries instead of nnode (a few KB that stay in L2 instead of a
  // megabyte per utterance written four bytes at a time).
  unsigned o_mp_tgt = 0;
  int n_mp_tgt = 0;
  std::vector<void *> owned;
};

struct jamd_beam {
  jamd_engine *eng = nullptr;
  jamd_lexicon *lex = nullptr;
  Work w{};
  int max_utts = 0;
  int *d_utt_off = nullptr;        // [nutt + 1] row offsets, then [nutt] the launch order (see upload_utt_off())
  std::vector<int> h_utt_off;      // host image of the same (the copy is asynchronous)
  bool strict = false;             // order mode JAMD_ORDER_STRICT
  bool exact = false;              // order mode JAMD_ORDER_EXACT (beam_exact.hip)
  int exact_status = -3;           // 0 = the exact-order kernel can serve this work area (xbeam_layout())
  XWork xw{};
  XWork xw_half{};                 // the same work area for the half shape (two workgroups per CU), when it fits
  int half_status = -2;            // 0 = xw_half is usable
  int shape_mode = JAMD_SHAPE_AUTO;
  bool stream_half = false;        // the shape of the open streaming session (the parked state is the layout's)
  hipEvent_t ev_started = nullptr; // recorded right before the latest first-pass launch (jamd_beam_wait_started())
  int *d_parr = nullptr;           // jamd_beam_prune_arrange(): the whole array
  unsigned *d_resident = nullptr;  // signal memory: first-pass workgroups started so far (Work::resident), nullptr = the device cannot wait on memory
  unsigned resident_target = 0;    // its value once the workgroups of the latest launch that fit the device at once have started
  unsigned launched_wg = 0;        // workgroups of all launches so far
  unsigned *d_pkeys = nullptr; int *d_pout = nullptr; size_t pcap = 0;   // jamd_beam_prune_order() scratch
  bool timed = false;               // JAMD_BEAM_TIMING=1: launch the instrumented instantiation
  int streaming = 0;               // utterances of the open streaming session, 0 = none
  int stream_pushes = 0;
  std::vector<int> stream_frames;  // frames pushed so far per utterance of the session (limit 32767 each)
  StrictWork sw{};
  std::vector<void *> owned;
};

// The exact-order kernel's workgroup shape for a launch of nutt utterances.  The full shape (1024 threads, a CU's whole
// LDS) is the faster one per utterance; the half shape lets two utterances share a CU, which pays once the batch has
// clearly more utterances than the device has CUs (one's barriers and wave-serial sections hide behind the other's work).
static bool use_half_shape(const jamd_beam *b, int nutt) {
  if (b->half_status != 0 || b->shape_mode == JAMD_SHAPE_FULL) return false;
  if (b->shape_mode == JAMD_SHAPE_HALF) return true;
  return nutt > b->eng->num_cu + b->eng->num_cu / 2;
}

// the frame-parallel (canonical tie) kernel: instantiation by where the survivor image lives and by instrumentation
static void launch_pass1(jamd_beam *b, const Work &w, int lds, int nutt, const float *dev_scores, int nstate, int smode,
                         hipStream_t st) {
  const dim3 grid(nutt), block(NT);
  if (w.use_lds) {
    if (b->timed) hipLaunchKernelGGL((beam_pass1_kernel<true, true>), grid, block, lds, st, b->lex->d, w, dev_scores, nstate, b->d_utt_off, smode);
    else hipLaunchKernelGGL((beam_pass1_kernel<false, true>), grid, block, lds, st, b->lex->d, w, dev_scores, nstate, b->d_utt_off, smode);
  } else {
    if (b->timed) hipLaunchKernelGGL((beam_pass1_kernel<true, false>), grid, block, lds, st, b->lex->d, w, dev_scores, nstate, b->d_utt_off, smode);
    else hipLaunchKernelGGL((beam_pass1_kernel<false, false>), grid, block, lds, st, b->lex->d, w, dev_scores, nstate, b->d_utt_off, smode);
  }
}

// Row offsets of the launch, followed by the ORDER in which the workgroups take the utterances: longest first.  With
// more utterances than CUs the dispatcher hands the next workgroup to the first CU that frees up, so longest-first is
// the classic greedy balance (a 512-utterance batch of 1 200-1 600-frame utterances: the slowest CU carries two average
// utterances instead of the two longest; 274 -> 245 ms).  The exact-order kernel reads it; the others ignore it.
static int upload_utt_off(jamd_beam *b, const int *utt_off, int nutt, hipStream_t st) {
  std::vector<int> &h = b->h_utt_off;
  h.assign((size_t)2 * nutt + 1, 0);
  for (int u = 0; u <= nutt; u++) h[(size_t)u] = utt_off[u];
  int *order = h.data() + nutt + 1;
  for (int u = 0; u < nutt; u++) order[u] = u;
  if (nutt > b->eng->num_cu)       // (one round: every workgroup starts at once, the order is irrelevant)
    std::stable_sort(order, order + nutt, [&](int a, int c) { return utt_off[a + 1] - utt_off[a] > utt_off[c + 1] - utt_off[c]; });
  JAMD_HIP(hipMemcpyAsync(b->d_utt_off, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice, st));
  return JAMD_OK;
}

// An event behind everything the launch stream holds before the first-pass kernel: it completes when that kernel is
// next to run (jamd_beam_wait_started()).  The workgroups of the launch are NOT accounted here: account_launch() does
// that once the launch is known to have been accepted, so a refused or failed call leaves the resident counter's
// bookkeeping where the device's counter will really be (a phantom workgroup would make every later
// jamd_beam_stream_wait_resident() wait for a value the counter never reaches).
static int mark_started(jamd_beam *b, hipStream_t st) {
  if (!b->ev_started) JAMD_HIP(hipEventCreateWithFlags(&b->ev_started, hipEventDisableTiming));
  // The counter and its targets are 32-bit and compared with >=: long before they could wrap (2^31 workgroups), drain
  // the device once and start again from zero.  (A reset enqueued on the launch stream would not do: a wait that another
  // stream has queued but not yet evaluated would then see 0 against its old target.)
  if (b->d_resident && b->launched_wg > 0x7fffffffu) {
    JAMD_HIP(hipDeviceSynchronize());
    JAMD_HIP(hipMemset(b->d_resident, 0, sizeof(unsigned)));
    b->launched_wg = 0; b->resident_target = 0;
  }
  JAMD_HIP(hipEventRecord(b->ev_started, st));
  return JAMD_OK;
}

// After a launch that hipGetLastError() accepted.  `counted`: its workgroups bump Work::resident when they start.
static void account_launch(jamd_beam *b, int nutt, bool counted) {
  if (counted) {
    // what fits the device at once: one workgroup per CU, two in the exact-order kernel's half shape
    const int cap = b->eng->num_cu * ((b->exact && !b->strict && use_half_shape(b, nutt)) ? 2 : 1);
    // ... less a sixteenth: a launch that fills the device has nearly all of its workgroups placed within microseconds,
    // but the last handful may start only when others end (measured on 512 utterances: any threshold up to 98 % releases
    // the waiting stream at once, 100 % holds it for 170 ms; JAMD_RESIDENT_SHARE=<percent> for experiments)
    int share = nutt < cap ? nutt : cap;
    int pct = 94;
    { const char *pc = getenv("JAMD_RESIDENT_SHARE"); if (pc && atoi(pc) > 0 && atoi(pc) <= 100) pct = atoi(pc); }
    share = (int)((long long)share * pct / 100);
    b->resident_target = b->launched_wg + (unsigned)(share > 0 ? share : 1);
    b->launched_wg += (unsigned)nutt;
  } else b->resident_target = b->launched_wg;
}

extern "C" {

int jamd_lexicon_create(jamd_engine *e, const jamd_lexicon_desc *h, jamd_lexicon **out) {
  if (!e || !h || !out) { jamd_set_error("jamd_lexicon_create: NULL argument"); return JAMD_EINVAL; }
  *out = nullptr;
  const int lmt = h->lm_type & 0xff;
  const bool multipath = (h->lm_type & JAMD_LM_MULTIPATH) != 0;
  const bool wordmode = lmt == JAMD_LM_WORD;
  const bool dfa = lmt == JAMD_LM_DFA || wordmode;             // the two LM_DFA variants share everything but the word boundary
  if ((h->lm_type & ~(0xff | JAMD_LM_MULTIPATH)) != 0 || (lmt != JAMD_LM_NGRAM && !dfa)) {
    jamd_set_error("jamd_lexicon_create: lm_type=%d", h->lm_type); return JAMD_EINVAL;
  }
  if (dfa && (h->ninit < 0 || (h->ninit > 0 && (!h->init_node || !h->init_lscore)) ||
              (!wordmode && (h->ncat <= 0 || !h->cat_pair || !h->start2wid)))) {
    jamd_set_error("jamd_lexicon_create: grammar descriptor incomplete (ncat=%d ninit=%d)", h->ncat, h->ninit);
    return JAMD_EINVAL;
  }
  if (h->nnode <= 0 || h->nword <= 0 || h->startnum < 0 || (!dfa && (h->head_silwid < 0 || h->head_silwid >= h->nword))) {
    jamd_set_error("jamd_lexicon_create: bad sizes (nnode=%d nword=%d head_silwid=%d)", h->nnode, h->nword,
                   h->head_silwid);
    return JAMD_EINVAL;
  }
  if (h->cdset_method == JAMD_IWCD_NBEST && (h->cdmax_num < 1 || h->cdmax_num > jamd::kNbestMax)) {
    jamd_set_error("jamd_lexicon_create: cdmax_num=%d outside [1,%d]", h->cdmax_num, jamd::kNbestMax);
    return JAMD_EINVAL;
  }
  // every index a kernel will follow is range-checked here: a truncated or corrupt blob (jamd_lexicon_load)
  // must fail with JAMD_EINVAL, not read out of bounds on the host or the device
  if (!h->ac_off || !h->self_a || !h->next_a || !h->stend || !h->scid || !h->out_id || !h->out_kind ||
      (h->startnum > 0 && !h->startnode) || !h->word_head || !h->wton || h->ac_off[0] != 0) {
    jamd_set_error("jamd_lexicon_create: NULL or malformed node arrays"); return JAMD_EINVAL;
  }
  int maxfan = 2;
  for (int i = 0; i < h->nnode; i++) {
    const int x = h->ac_off[i + 1] - h->ac_off[i];
    if (x < 0) { jamd_set_error("jamd_lexicon_create: ac_off not monotone at node %d", i); return JAMD_EINVAL; }
    if (2 + x > maxfan) maxfan = 2 + x;
    if (h->stend[i] >= h->nword) { jamd_set_error("jamd_lexicon_create: node %d ends word %d of %d", i, h->stend[i], h->nword); return JAMD_EINVAL; }
    if (h->scid[i] >= h->nscword || (h->scid[i] < 0 && -h->scid[i] >= h->nfscore)) {
      jamd_set_error("jamd_lexicon_create: node %d has successor id %d outside the tables", i, h->scid[i]); return JAMD_EINVAL;
    }
  }
  for (int k = 0; k < h->ac_off[h->nnode]; k++)
    if (h->ac_to[k] < 0 || h->ac_to[k] >= h->nnode) { jamd_set_error("jamd_lexicon_create: arc %d leads to node %d of %d", k, h->ac_to[k], h->nnode); return JAMD_EINVAL; }
  for (int s = 0; s < h->startnum; s++)
    if (h->startnode[s] < 0 || h->startnode[s] >= h->nnode) { jamd_set_error("jamd_lexicon_create: root %d is node %d of %d", s, h->startnode[s], h->nnode); return JAMD_EINVAL; }
  for (int w = 0; w < h->nword; w++)
    if (h->word_head[w] < -1 || h->word_head[w] >= h->nnode) { jamd_set_error("jamd_lexicon_create: word %d starts at node %d of %d", w, h->word_head[w], h->nnode); return JAMD_EINVAL; }
  if (h->nset > 0) {
    if (!h->set_off || !h->set_states || h->set_off[0] != 0) { jamd_set_error("jamd_lexicon_create: state-set table missing"); return JAMD_EINVAL; }
    for (int i = 0; i < h->nset; i++)
      if (h->set_off[i + 1] < h->set_off[i]) { jamd_set_error("jamd_lexicon_create: set_off not monotone at %d", i); return JAMD_EINVAL; }
    for (int k = 0; k < h->set_off[h->nset]; k++)
      if (h->set_states[k] < 0) { jamd_set_error("jamd_lexicon_create: negative state in set table"); return JAMD_EINVAL; }
  }
  std::vector<int> iso(h->isolatenum > 0 ? h->isolatenum : 0, -1), shared;
  for (int s = 0; s < h->startnum && !dfa; s++) {
    const int i = h->start2isolate[s];
    if (i >= 0) {
      if (i >= h->isolatenum || iso[i] >= 0) { jamd_set_error("jamd_lexicon_create: start2isolate out of range or repeated"); return JAMD_EINVAL; }
      iso[i] = s;
      const int sc = h->scid[h->startnode[s]];
      if (sc <= 0 || sc >= h->nscword) { jamd_set_error("jamd_lexicon_create: isolated root without a successor word"); return JAMD_EINVAL; }
    } else {
      const int sc = h->scid[h->startnode[s]];
      if (sc >= 0 || -sc >= h->nfscore) { jamd_set_error("jamd_lexicon_create: shared root without a factoring value"); return JAMD_EINVAL; }
      shared.push_back(s);
    }
  }
  if (h->nword >= (1 << 30)) { jamd_set_error("jamd_lexicon_create: nword=%d too large", h->nword); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  jamd_lexicon *l = new jamd_lexicon();
  l->eng = e; l->maxfan = maxfan; l->nscword = h->nscword; l->multipath = multipath;
  if (multipath) {
    // A root that reaches a word-end node along its own arcs (a word of tee models only): a cross-word transition would
    // improve a word end inside the loop that visits the word ends (beam.c:2779-2825), and the result depends on the loop's
    // position -- strict order only.  Roots are non-emitting; so is every word end of a multipath lexicon.
    bool ok = true;
    for (int s = 0; s < h->startnum && ok; s++) {
      const int r = h->startnode[s];
      if (h->self_a[r] != JAMD_LOG_ZERO && h->stend[r] >= 0) ok = false;
      if (h->next_a[r] != JAMD_LOG_ZERO && r + 1 < h->nnode && h->stend[r + 1] >= 0) ok = false;
      for (int k = h->ac_off[r]; k < h->ac_off[r + 1]; k++) if (h->stend[h->ac_to[k]] >= 0) ok = false;
    }
    l->mp_parallel = ok;
  }
  LexDev &d = l->d;
  d.nnode = h->nnode; d.nword = h->nword; d.startnum = h->startnum; d.isolatenum = h->isolatenum;
  d.nshared = (int)shared.size(); d.nlc = h->nlc; d.cdset_method = h->cdset_method; d.cdmax_num = h->cdmax_num;
  d.head_silwid = h->head_silwid; d.tail_silwid = h->tail_silwid; d.ng_mode = h->ng_mode; d.ng_unk_id = h->ng_unk_id;
  d.ng_unk_num_log = h->ng_unk_num_log; d.lm_weight = h->lm_weight; d.lm_penalty = h->lm_penalty;
  d.lm_penalty_trans = h->lm_penalty_trans;
  const int nac = h->ac_off[h->nnode], nset_states = h->nset ? h->set_off[h->nset] : 0;
  int rc = JAMD_OK;
  // every array is appended (16-byte aligned) to one host image that is uploaded once; the kernel
  // sees the base pointer and 32-bit byte offsets (see LexDev)
  std::vector<unsigned char> arena;
#define UP(field, src, n)                                                              \
  do {                                                                                 \
    const size_t bytes_ = sizeof(*(src)) * (size_t)(n), at_ = (arena.size() + 15) & ~(size_t)15;   \
    arena.resize(at_ + (bytes_ ? bytes_ : 16));                                        \
    if (bytes_) memcpy(arena.data() + at_, (src), bytes_);                             \
    d.o_##field = (unsigned)at_;                                                       \
  } while (0)
  std::vector<int4> na(h->nnode), nb(h->nnode);
  std::vector<int> word_end(h->nword, -1);
  for (int i = 0; i < h->nnode; i++) {
    int sa, nx;
    memcpy(&sa, &h->self_a[i], 4); memcpy(&nx, &h->next_a[i], 4);
    na[i] = make_int4(sa, nx, h->ac_off[i], h->ac_off[i + 1]);
    nb[i] = make_int4(h->stend[i], h->scid[i], h->out_id[i], (int)h->out_kind[i]);
    if (h->stend[i] >= 0 && h->stend[i] < h->nword) word_end[h->stend[i]] = i;
  }
  // both root lists in the order beam_inter_word() / beam_inter_word_factoring() visit them (stid from
  // startnum-1 down to 0, beam.c:2334 / :2562): the exact-order kernel numbers its candidates by list index
  for (size_t i = 0; i < iso.size() && !dfa; i++)
    if (iso[i] < 0) { jamd_set_error("jamd_lexicon_create: isolated root %d is not assigned", (int)i); return JAMD_EINVAL; }
  std::sort(iso.begin(), iso.end(), [](int a, int b) { return a > b; });
  std::sort(shared.begin(), shared.end(), [](int a, int b) { return a > b; });
  std::vector<int2> iso_root(iso.size());
  for (size_t i = 0; i < iso.size(); i++) {
    const int node = h->startnode[iso[i]];
    iso_root[i] = make_int2(node, h->scword[h->scid[node]]);
  }
  std::vector<float2> shared_root(shared.size());
  for (size_t i = 0; i < shared.size(); i++) {
    const int node = h->startnode[shared[i]];
    float nf; memcpy(&nf, &node, 4);
    shared_root[i] = make_float2(nf, h->fscore[-h->scid[node]]);
  }
  {
    std::vector<int4> nab(2 * (size_t)h->nnode);       // one 32-byte record per node (LexDev::node_a / node_b / scid)
    for (int i = 0; i < h->nnode; i++) { nab[2 * (size_t)i] = na[i]; nab[2 * (size_t)i + 1] = nb[i]; }
    UP(node_a, nab.data(), nab.size());
    d.o_node_b = d.o_node_a + 16u; d.o_scid = d.o_node_a + 20u;
  }
  UP(ac_to, h->ac_to, nac); UP(ac_a, h->ac_a, nac);
  UP(iso_root, iso_root.data(), iso_root.size()); UP(shared_root, shared_root.data(), shared_root.size());
  UP(word_end, word_end.data(), word_end.size());
  if (multipath) {
    // the roots' own transitions, flattened once (csrc/beam_exact_mp.h, step B')
    auto expand = [&](int root, int rootno, int tag, std::vector<int4> &out) {
      auto bits = [](float f) { int b; memcpy(&b, &f, 4); return b; };
      if (h->self_a[root] != JAMD_LOG_ZERO) out.push_back(make_int4(root, bits(h->self_a[root]), rootno * maxfan + 0, tag));
      if (h->next_a[root] != JAMD_LOG_ZERO) out.push_back(make_int4(root + 1, bits(h->next_a[root]), rootno * maxfan + 1, tag));
      for (int k = h->ac_off[root]; k < h->ac_off[root + 1]; k++)
        out.push_back(make_int4(h->ac_to[k], bits(h->ac_a[k]), rootno * maxfan + 2 + (k - h->ac_off[root]), tag));
    };
    std::vector<int4> e_iso, e_shared, e_start;
    const int head_root = dfa ? -1 : h->word_head[h->head_silwid];
    for (size_t i = 0; i < iso_root.size() && !dfa; i++)
      if (iso_root[i].x != head_root) expand(iso_root[i].x, (int)i, (int)i, e_iso);                       // :2336-2341
    for (size_t r = 0; r < shared_root.size() && !dfa; r++) {
      int node; memcpy(&node, &shared_root[r].x, 4);
      int fs; memcpy(&fs, &shared_root[r].y, 4);
      if (node != head_root) expand(node, (int)r, fs, e_shared);                                          // :2566-2571
    }
    for (int rv = 0; rv < h->startnum && dfa && !wordmode; rv++) {
      const int r = h->startnum - 1 - rv;                                                                 // roots from startnum-1 down (:2334)
      expand(h->startnode[r], rv, r, e_start);
    }
    auto put = [&](const std::vector<int4> &v, unsigned *off, int *cnt) {
      const size_t at = (arena.size() + 15) & ~(size_t)15;
      arena.resize(at + (v.empty() ? 16 : v.size() * sizeof(int4)));
      if (!v.empty()) memcpy(arena.data() + at, v.data(), v.size() * sizeof(int4));
      *off = (unsigned)at; *cnt = (int)v.size();
    };
    put(e_iso, &l->o_mp_iso, &l->n_mp_iso); put(e_shared, &l->o_mp_shared, &l->n_mp_shared); put(e_start, &l->o_mp_start, &l->n_mp_start);
    {
      std::vector<int> tgt((size_t)h->nnode, -1);
      int ntgt = 0;
      for (const std::vector<int4> *v : {&e_iso, &e_shared, &e_start})
        for (const int4 &ent : *v) if (tgt[(size_t)ent.x] < 0) tgt[(size_t)ent.x] = ntgt++;
      const size_t at = (arena.size() + 15) & ~(size_t)15;
      arena.resize(at + tgt.size() * sizeof(int));
      memcpy(arena.data() + at, tgt.data(), tgt.size() * sizeof(int));
      l->o_mp_tgt = (unsigned)at; l->n_mp_tgt = ntgt;
    }
  }
  UP(startnode, h->startnode, h->startnum); UP(start2isolate, h->start2isolate, h->startnum);
  UP(lc_tab, h->lc_tab, (size_t)h->nlcrow * (h->nlc + 1)); UP(word_lc, h->word_lc, h->nword);
  UP(set_off, h->set_off, h->nset + 1); UP(set_states, h->set_states, nset_states);
  UP(wordend_a, h->wordend_a, h->nword); UP(wton, h->wton, h->nword); UP(cprob, h->cprob, h->nword);
  UP(is_transparent, h->is_transparent, h->nword); UP(word_head, h->word_head, h->nword);
  UP(fscore, h->fscore, h->nfscore); UP(scword, h->scword, h->nscword);
  UP(ng_uni_prob, h->ng_uni_prob, h->ng_nword); UP(ng_uni_bo, h->ng_uni_bo, h->ng_nword);
  UP(ng_bi_bgn, h->ng_bi_bgn, h->ng_nword); UP(ng_bi_num, h->ng_bi_num, h->ng_nword);
  UP(ng_bi_wid, h->ng_bi_wid, h->ng_nbigram); UP(ng_bi_prob, h->ng_bi_prob, h->ng_nbigram);
  d.lm_type = lmt; d.ncat = dfa ? h->ncat : 0; d.ninit = dfa ? h->ninit : 0; d.penalty1 = dfa ? h->penalty1 : 0.0f;
  if (dfa) {
    std::vector<int> root_cat(h->startnum, 0);
    for (int s = 0; s < h->startnum && !wordmode; s++) {
      const int w = h->start2wid[s];
      if (w < 0 || w >= h->nword || h->wton[w] < 0 || h->wton[w] >= h->ncat) {
        jamd_set_error("jamd_lexicon_create: root %d has no valid category", s); rc = JAMD_EINVAL; break;
      }
      root_cat[s] = h->wton[w];
    }
    for (int w = 0; w < h->nword && rc == JAMD_OK && !wordmode; w++)
      if (h->wton[w] < 0 || h->wton[w] >= h->ncat) { jamd_set_error("jamd_lexicon_create: word %d outside the categories", w); rc = JAMD_EINVAL; }
    for (int e = 0; e < h->ninit && rc == JAMD_OK; e++)
      if (h->init_node[e] < 0 || h->init_node[e] >= h->nnode) { jamd_set_error("jamd_lexicon_create: bad initial node"); rc = JAMD_EINVAL; }
    UP(cat_pair, h->cat_pair, wordmode ? 0 : (size_t)h->ncat * h->ncat); UP(root_cat, root_cat.data(), root_cat.size());
    UP(init_node, h->init_node, h->ninit); UP(init_lscore, h->init_lscore, h->ninit);
    if (h->nfwd > 0 && rc == JAMD_OK) {
      // forward DFA: every index the kernels will follow is checked here
      if (wordmode || !h->fwd_off || !h->fwd_label || !h->fwd_to || !h->init_to_state || h->fwd_off[0] != 0) {
        jamd_set_error("jamd_lexicon_create: forward DFA descriptor incomplete"); rc = JAMD_EINVAL;
      }
      for (int s2 = 0; s2 < h->nfwd && rc == JAMD_OK; s2++)
        if (h->fwd_off[s2 + 1] < h->fwd_off[s2]) { jamd_set_error("jamd_lexicon_create: forward DFA offsets not monotone"); rc = JAMD_EINVAL; }
      for (int a = 0; rc == JAMD_OK && a < h->fwd_off[h->nfwd]; a++)
        if (h->fwd_to[a] < 0 || h->fwd_to[a] >= h->nfwd) { jamd_set_error("jamd_lexicon_create: forward DFA arc %d leaves the automaton", a); rc = JAMD_EINVAL; }
      for (int e2 = 0; rc == JAMD_OK && e2 < h->ninit; e2++)
        if (h->init_to_state[e2] < -1 || h->init_to_state[e2] >= h->nfwd) { jamd_set_error("jamd_lexicon_create: bad initial forward-DFA state"); rc = JAMD_EINVAL; }
      if (rc == JAMD_OK) {
        UP(fwd_off, h->fwd_off, (size_t)h->nfwd + 1); UP(fwd_label, h->fwd_label, (size_t)h->fwd_off[h->nfwd]);
        UP(fwd_to, h->fwd_to, (size_t)h->fwd_off[h->nfwd]); UP(init_to_state, h->init_to_state, h->ninit);
        d.nfwd = h->nfwd;
      }
    }
  }
#undef UP
  if (rc == JAMD_OK && arena.size() >= ((size_t)1 << 32)) { jamd_set_error("jamd_lexicon_create: lexicon image exceeds 4 GB"); rc = JAMD_EINVAL; }
  if (rc == JAMD_OK) {
    unsigned char *dev = nullptr;
    rc = upload(&dev, arena.data(), arena.size());
    d.base = dev;
    if (dev) l->owned.push_back((void *)dev);
  }
  if (rc == JAMD_OK && !dfa && h->isolatenum > 0 && h->ng_nword > 0 &&
      (size_t)h->ng_nword * h->isolatenum * sizeof(float) <= kIwTabMaxBytes) {
    const size_t cells = (size_t)h->ng_nword * h->isolatenum;
    float *tab = nullptr;
    if (hipMalloc((void **)&tab, cells * sizeof(float)) == hipSuccess) {
      l->owned.push_back((void *)tab);
      hipLaunchKernelGGL(iwtab_build_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, e->stream, d, tab, h->ng_nword, h->isolatenum);
      if (hipGetLastError() == hipSuccess && hipStreamSynchronize(e->stream) == hipSuccess) d.iwtab = tab;
    } else (void)hipGetLastError();             // no room: the kernels compute the entries on the fly
  }
  if (rc != JAMD_OK) { jamd_lexicon_destroy(l); return rc; }
  *out = l;
  return JAMD_OK;
}

void jamd_lexicon_destroy(jamd_lexicon *l) {
  if (!l) return;
  (void)hipSetDevice(l->eng->device);
  for (void *p : l->owned) (void)hipFree(p);
  delete l;
}

int jamd_beam_create(jamd_engine *e, jamd_lexicon *l, int beam_width, float score_pruning_width,
                     int max_utts, int atoms_per_utt, jamd_beam **out) {
  if (!e || !l || !out) { jamd_set_error("jamd_beam_create: NULL argument"); return JAMD_EINVAL; }
  *out = nullptr;
  if (beam_width < 1 || beam_width > 65536) {
    jamd_set_error("jamd_beam_create: beam_width=%d outside [1,65536]", beam_width);
    return JAMD_EINVAL;
  }
  if (max_utts < 1 || atoms_per_utt < 1) { jamd_set_error("jamd_beam_create: bad capacity"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  jamd_beam *b = new jamd_beam();
  b->eng = e; b->lex = l; b->max_utts = max_utts;
  { const char *tm = getenv("JAMD_BEAM_TIMING"); b->timed = tm != nullptr && atoi(tm) != 0; }
  Work &w = b->w;
  w.beam = beam_width; w.width = score_pruning_width; w.nnode = l->d.nnode; w.nword = l->d.nword;
  w.atom_cap = atoms_per_utt;
  // every survivor reaches at most maxfan nodes, cross-word candidates only reach roots (multipath: what the roots reach)
  w.tok_cap = beam_width * l->maxfan + l->d.startnum * (l->multipath ? l->maxfan : 1) + l->d.ninit + 1;
  const size_t U = (size_t)max_utts;
  int rc = JAMD_OK;
  auto alloc = [&](void **p, size_t bytes, bool zero) -> int {
    JAMD_HIP(hipMalloc(p, bytes ? bytes : 4));
    b->owned.push_back(*p);
    if (zero) JAMD_HIP(hipMemset(*p, 0, bytes));
    return JAMD_OK;
  };
  // survivor state: Tok[beam] + atom[beam] + welist[beam] + hash keys/values[hsize]
  w.hsize = 64; while (w.hsize < 2 * beam_width) w.hsize <<= 1;
  w.sv_bytes = (int)(beam_width * (sizeof(Tok) + 2 * sizeof(int)) + (size_t)w.hsize * 2 * sizeof(int));
  w.sv_bytes = (w.sv_bytes + 15) & ~15;
  w.use_lds = w.sv_bytes + kHistBytes <= kMaxDynLds ? 1 : 0;
  // what is left of the LDS budget holds the frame's Viterbi cells (12 bytes per slot), if that is
  // at least 4096 slots; a frame that outgrows the table overflows into nodekey[]
  w.cell_slots = 0;
  if (w.use_lds) {
    int slots = 4096;
    while ((size_t)w.sv_bytes + (size_t)slots * 2 * 12 <= (size_t)kMaxDynLds && slots < 65536) slots *= 2;
    if ((size_t)w.sv_bytes + (size_t)slots * 12 <= (size_t)kMaxDynLds) w.cell_slots = slots;
    // a frame creates about five tokens per survivor; a table that would run above ~2/3 load
    // costs more in failed probes than it saves, so narrower-than-needed tables are not used
    if (w.cell_slots < 8 * beam_width) w.cell_slots = 0;
#ifdef JAMD_DEV
    if (getenv("JAMD_BEAM_NO_LDS_CELLS") != nullptr) w.cell_slots = 0;      // development switch (timing comparison)
#endif
  }
  w.cell_off = w.use_lds ? w.sv_bytes : 0;
  w.node_off = w.cell_off + (8 * w.cell_slots > kHistBytes ? 8 * w.cell_slots : kHistBytes);
  w.row_off = w.node_off + 4 * w.cell_slots;
  w.lds_bytes = w.row_off;
  w.row_cache = 0;
  // one slice per utterance: every array at a 256-byte aligned 32-bit offset
  w.nscword = l->nscword > 0 ? l->nscword : 1;
  {
    size_t at = 0;
    auto place = [&](unsigned *off, size_t bytes) { *off = (unsigned)at; at = (at + bytes + 255) & ~(size_t)255; };
    place(&w.o_nodekey, (size_t)w.nnode * sizeof(unsigned long long));
    place(&w.o_cur, (size_t)w.tok_cap * (sizeof(Tok) + 16));   // + the exact-order kernel's 16-byte records of a frame's tokens (REC())
    place(&w.o_cur_key, (size_t)w.tok_cap * sizeof(unsigned));
    place(&w.o_touched, (size_t)w.tok_cap * sizeof(int2));
    place(&w.o_arcq, (size_t)w.tok_cap * sizeof(int2));
    place(&w.o_atoms, (size_t)w.atom_cap * sizeof(jamd_trellis_atom));
    place(&w.o_lmcache, (size_t)w.nscword * sizeof(unsigned long long));
    // exact-order kernel (beam_exact.hip): its LDS layout, and its three extra per-utterance arrays
    XWork &xw = b->xw;
    const bool mp = l->multipath;
    const int mp_roots = (l->d.lm_type == JAMD_LM_NGRAM) ? l->d.isolatenum : l->d.startnum;   // roots a word end is followed by
    b->exact_status = (mp && !l->mp_parallel) ? -4
                      : xbeam_layout(&xw, w, l->maxfan, mp ? mp_roots : l->d.startnum, l->d.ninit, l->d.nshared, false, mp);
    b->half_status = b->exact_status != 0 ? -2
                     : xbeam_layout(&b->xw_half, w, l->maxfan, mp ? mp_roots : l->d.startnum, l->d.ninit, l->d.nshared, true, mp);
    if (mp && getenv("JAMD_MP_HALF_OFF") != nullptr) b->half_status = -2;    // (development: the multipath frame in the full shape only, as in round 4)
    size_t sv_max = (size_t)w.sv_bytes;
    if (b->exact_status == 0 && (size_t)xw.w.sv_bytes > sv_max) sv_max = (size_t)xw.w.sv_bytes;
    if (b->half_status == 0 && (size_t)b->xw_half.w.sv_bytes > sv_max) sv_max = (size_t)b->xw_half.w.sv_bytes;
    place(&w.o_sv, sv_max);
    if (b->exact_status == 0) {
      // the bitmap holds one bit per visiting index: maxfan per survivor plus startnum per word end
      size_t bits = (size_t)(beam_width + 2) * (size_t)(l->maxfan + l->d.startnum) + (size_t)l->d.nshared + (size_t)l->d.ninit + 64;
      if (mp) bits = (size_t)(beam_width + 2) * (size_t)l->maxfan * (size_t)(mp_roots > 1 ? mp_roots : 1) + (size_t)l->d.nshared * l->maxfan + 64;
      place(&xw.o_nodefirst, (size_t)w.nnode * sizeof(unsigned));
      place(&xw.o_bitmap, (bits + 31) / 32 * 4);
      place(&xw.o_heap, ((size_t)w.tok_cap + 2) * sizeof(unsigned long long));
      place(&xw.o_collect, ((size_t)beam_width + 256) * 16);
      place(&xw.o_sweep, xbeam_sweep_bytes(beam_width));
      place(&xw.o_pstat, 16 * sizeof(int));
      xw.o_nodetok = xw.o_arr = xw.o_key2 = 0;
      xw.o_mp_iso = l->o_mp_iso; xw.o_mp_shared = l->o_mp_shared; xw.o_mp_start = l->o_mp_start;
      xw.n_mp_iso = l->n_mp_iso; xw.n_mp_shared = l->n_mp_shared; xw.n_mp_start = l->n_mp_start;
      xw.o_mp_tgt = l->o_mp_tgt; xw.n_mp_tgt = l->n_mp_tgt;
      if (mp) {
        place(&xw.o_nodetok, (size_t)(l->n_mp_tgt > 0 ? l->n_mp_tgt : 1) * sizeof(unsigned));
        place(&xw.o_arr, (size_t)w.tok_cap * sizeof(int));
        place(&xw.o_key2, (size_t)w.tok_cap * sizeof(unsigned));
      }
    }
    if (at >= ((size_t)1 << 32)) { jamd_set_error("jamd_beam_create: per-utterance work area exceeds 4 GB"); rc = JAMD_EINVAL; }
    w.utt_stride = at;
  }
  if (rc == JAMD_OK) rc = alloc((void **)&w.slices, U * (size_t)w.utt_stride, true);    // zero: empty Viterbi cells
  if (rc == JAMD_OK) rc = alloc((void **)&w.res, U * sizeof(jamd_pass1_result), true);
  w.resident = nullptr;
  if (rc == JAMD_OK) {
    // a counter the first-pass workgroups bump when they start, in signal memory so that another stream's command
    // processor can wait on it (jamd_beam_stream_wait_resident()); JAMD_NO_WAIT_VALUE=1 keeps the host-side wait
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, e->device);
    const char *off = getenv("JAMD_NO_WAIT_VALUE");
    if (can && !(off && off[0] == '1')) {
      void *p = nullptr;
      if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) == hipSuccess && p) {
        if (hipMemset(p, 0, 8) == hipSuccess) { b->d_resident = (unsigned *)p; w.resident = b->d_resident; }
        else (void)hipFree(p);
      }
      (void)hipGetLastError();
    }
  }
  if (rc == JAMD_OK) {
    // the attribute is per kernel, not per work area: always ask for the whole budget
    hipError_t ae = hipFuncSetAttribute((const void *)beam_pass1_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        kMaxDynLds);
    if (ae == hipSuccess)
      ae = hipFuncSetAttribute((const void *)beam_pass1_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    if (ae == hipSuccess)
      ae = hipFuncSetAttribute((const void *)beam_pass1_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    if (ae == hipSuccess)
      ae = hipFuncSetAttribute((const void *)beam_pass1_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    if (ae != hipSuccess) { jamd_set_error("jamd_beam_create: cannot reserve %d bytes of LDS: %s", w.sv_bytes,
                                           hipGetErrorString(ae)); rc = JAMD_ENODEV; }
  }
  if (rc == JAMD_OK) rc = alloc((void **)&b->d_utt_off, (2 * U + 1) * sizeof(int), true);
  if (rc == JAMD_OK && b->exact_status == 0) {
    // same slices, same offsets; only the LDS image differs
    const int svb = b->xw.w.sv_bytes;
    b->xw.w = w; b->xw.w.sv_bytes = svb;
    if (b->half_status == 0) {
      XWork &xh = b->xw_half;
      const int svh = xh.w.sv_bytes;
      xh.w = w; xh.w.sv_bytes = svh;
      xh.o_nodefirst = b->xw.o_nodefirst; xh.o_bitmap = b->xw.o_bitmap; xh.o_heap = b->xw.o_heap; xh.o_collect = b->xw.o_collect; xh.o_sweep = b->xw.o_sweep; xh.o_pstat = b->xw.o_pstat;
      xh.o_nodetok = b->xw.o_nodetok; xh.o_arr = b->xw.o_arr; xh.o_key2 = b->xw.o_key2;
      xh.o_mp_iso = b->xw.o_mp_iso; xh.o_mp_shared = b->xw.o_mp_shared; xh.o_mp_start = b->xw.o_mp_start;
      xh.n_mp_iso = b->xw.n_mp_iso; xh.n_mp_shared = b->xw.n_mp_shared; xh.n_mp_start = b->xw.n_mp_start;
      xh.o_mp_tgt = b->xw.o_mp_tgt; xh.n_mp_tgt = b->xw.n_mp_tgt;
    }
    if (xbeam_prepare() != hipSuccess) b->exact_status = -5;
  }
  // default order mode: the exact-order kernel where it can serve the work area, else the frame-parallel one
  b->exact = b->exact_status == 0;
  if (rc == JAMD_OK && l->d.nfwd > 0 && !b->exact) {
    // The canonical-tie kernel carries no forward-DFA state; the strict-order kernels do.  A work area the exact-order
    // kernel cannot serve (beam too wide for the LDS image, a root that reaches a word end, no LDS) therefore starts in
    // strict order instead of being refused: the caller has no beam to call jamd_beam_set_strict_order() on
    // when create fails.
    rc = jamd_beam_set_strict_order(b, 1);
    if (rc != JAMD_OK) jamd_set_error("jamd_beam_create: a grammar with a forward DFA needs the exact-order or the strict-order kernel; "
                                      "neither can serve beam %d on this lexicon", w.beam);
  }
  if (rc != JAMD_OK) { jamd_beam_destroy(b); return rc; }
  *out = b;
  return JAMD_OK;
}

void jamd_beam_destroy(jamd_beam *b) {
  if (!b) return;
  (void)hipSetDevice(b->eng->device);
  for (void *p : b->owned) (void)hipFree(p);
  if (b->ev_started) (void)hipEventDestroy(b->ev_started);
  if (b->d_resident) (void)hipFree(b->d_resident);
  delete b;
}

int jamd_beam_pass1_dev(jamd_beam *b, const float *dev_scores, int nstate, const int *utt_off, int nutt,
                        void *stream) {
  if (!b || !dev_scores || !utt_off || nstate <= 0) { jamd_set_error("jamd_beam_pass1_dev: bad argument"); return JAMD_EINVAL; }
  if (nutt < 0 || nutt > b->max_utts) {
    jamd_set_error("jamd_beam_pass1_dev: nutt=%d exceeds the work area (%d)", nutt, b->max_utts);
    return JAMD_EINVAL;
  }
  for (int u = 0; u < nutt; u++) {
    const int T = utt_off[u + 1] - utt_off[u];
    if (T < 0 || T > 32767) {   // TRELLIS_ATOM times are short (trellis.h:32-33)
      jamd_set_error("jamd_beam_pass1_dev: utterance %d has %d frames (limit 32767)", u, T);
      return JAMD_EINVAL;
    }
  }
  if (nutt == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(b->eng->device));
  hipStream_t st = jamd_stream(b->eng, stream);
  if (b->lex->multipath && !b->strict && !b->exact) {   // (state checks come before anything is enqueued or accounted)
    jamd_set_error("jamd_beam_pass1_dev: this multipath lexicon is decoded by the strict-order kernel only: "
                   "jamd_beam_set_strict_order(b, 1)");
    return JAMD_ESTATE;
  }
  { const int rc = upload_utt_off(b, utt_off, nutt, st); if (rc != JAMD_OK) return rc; }
  { const int rc = mark_started(b, st); if (rc != JAMD_OK) return rc; }
  if (b->strict && b->lex->multipath)
    hipLaunchKernelGGL(beam_strict_mp_kernel, dim3((nutt + 63) / 64), dim3(64), 0, st, b->lex->d, b->w, b->sw, dev_scores,
                       nstate, b->d_utt_off, nutt);
  else if (b->strict)
    hipLaunchKernelGGL(beam_strict_kernel, dim3((nutt + 63) / 64), dim3(64), 0, st, b->lex->d, b->w, b->sw, dev_scores,
                       nstate, b->d_utt_off, nutt);
  else if (b->exact)
    xbeam_launch(b->lex->d, use_half_shape(b, nutt) ? b->xw_half : b->xw, dev_scores, nstate, b->d_utt_off, nutt, 0, b->timed, st);
  else {
    Work w = b->w;                                     // the score row joins the LDS image when it still fits
    w.row_cache = w.lds_bytes + 4 * nstate <= kMaxDynLds;
#ifdef JAMD_DEV
    if (getenv("JAMD_BEAM_NO_ROW_CACHE") != nullptr) w.row_cache = 0;
#endif
    const int lds = w.lds_bytes + (w.row_cache ? 4 * nstate : 0);
    launch_pass1(b, w, lds, nutt, dev_scores, nstate, 0, st);
  }
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { jamd_set_error("jamd_beam_pass1_dev: launch failed: %s", hipGetErrorString(le)); return JAMD_ELAUNCH; }
  account_launch(b, nutt, !b->strict);
  return JAMD_OK;
}

int jamd_beam_stream_begin(jamd_beam *b, int nutt) {
  if (!b || nutt < 1 || nutt > b->max_utts) { jamd_set_error("jamd_beam_stream_begin: bad argument"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(b->eng->device));
  void *p = nullptr;
  if (b->w.stream == nullptr) {
    JAMD_HIP(hipMalloc(&p, sizeof(StreamState) * (size_t)b->max_utts)); b->owned.push_back(p); b->w.stream = (StreamState *)p;
  }
  JAMD_HIP(hipMemsetAsync(b->w.stream, 0, sizeof(StreamState) * (size_t)nutt, b->eng->stream));
  JAMD_HIP(hipStreamSynchronize(b->eng->stream));
  b->streaming = nutt; b->stream_pushes = 0;
  b->stream_half = use_half_shape(b, nutt);           // one shape for the whole session: the parked state is the layout's
  b->stream_frames.assign((size_t)nutt, 0);
  return JAMD_OK;
}

int jamd_beam_stream_push_dev(jamd_beam *b, const float *dev_scores, int nstate, const int *chunk_off, int nutt,
                              int final, void *stream) {
  if (!b || !chunk_off || nstate <= 0 || (!dev_scores && chunk_off[nutt > 0 ? nutt : 0] > 0)) {
    jamd_set_error("jamd_beam_stream_push_dev: bad argument"); return JAMD_EINVAL;
  }
  if (b->streaming <= 0 || nutt != b->streaming) {
    jamd_set_error("jamd_beam_stream_push_dev: call jamd_beam_stream_begin(b, %d) first", nutt); return JAMD_ESTATE;
  }
  for (int u = 0; u < nutt; u++) {
    if (chunk_off[u + 1] < chunk_off[u]) { jamd_set_error("jamd_beam_stream_push_dev: chunk_off must be non-decreasing"); return JAMD_EINVAL; }
    if ((long)b->stream_frames[u] + (chunk_off[u + 1] - chunk_off[u]) > 32767) {   // TRELLIS_ATOM times are short
      jamd_set_error("jamd_beam_stream_push_dev: utterance %d would exceed 32767 frames", u); return JAMD_EINVAL;
    }
  }
  if (b->lex->multipath && !b->strict && !b->exact) {
    jamd_set_error("jamd_beam_stream_push_dev: this multipath lexicon is decoded by the strict-order kernel only");
    return JAMD_ESTATE;
  }
  if (b->strict && (!final || b->stream_pushes != 0)) {
    // the strict-order kernel keeps no state between launches: one push carrying everything
    jamd_set_error("jamd_beam_stream_push_dev: strict-order mode needs the whole utterance in one final push");
    return JAMD_ESTATE;
  }
  for (int u = 0; u < nutt; u++) b->stream_frames[u] += chunk_off[u + 1] - chunk_off[u];   // only an accepted push counts
  if (b->strict) {
    b->streaming = 0;
    return jamd_beam_pass1_dev(b, dev_scores, nstate, chunk_off, nutt, stream);
  }
  b->stream_pushes++;
  JAMD_HIP(hipSetDevice(b->eng->device));
  hipStream_t st = jamd_stream(b->eng, stream);
  { const int rc = upload_utt_off(b, chunk_off, nutt, st); if (rc != JAMD_OK) return rc; }
  { const int rc = mark_started(b, st); if (rc != JAMD_OK) return rc; }
  if (b->exact) {
    b->xw.w.stream = b->w.stream; b->xw_half.w.stream = b->w.stream;
    xbeam_launch(b->lex->d, b->stream_half ? b->xw_half : b->xw, dev_scores, nstate, b->d_utt_off, nutt, final ? 2 : 1, b->timed, st);
  } else {
    Work w = b->w;
    w.row_cache = w.lds_bytes + 4 * nstate <= kMaxDynLds;
#ifdef JAMD_DEV
    if (getenv("JAMD_BEAM_NO_ROW_CACHE") != nullptr) w.row_cache = 0;
#endif
    const int lds = w.lds_bytes + (w.row_cache ? 4 * nstate : 0);
    launch_pass1(b, w, lds, nutt, dev_scores, nstate, final ? 2 : 1, st);
  }
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { jamd_set_error("jamd_beam_stream_push_dev: launch failed: %s", hipGetErrorString(le)); return JAMD_ELAUNCH; }
  account_launch(b, nutt, true);
  if (final) b->streaming = 0;
  return JAMD_OK;
}

int jamd_beam_set_strict_order(jamd_beam *b, int on) {
  if (!b) { jamd_set_error("jamd_beam_set_strict_order: NULL"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(b->eng->device));
  if (on && b->sw.token == nullptr) {
    const size_t U = (size_t)b->max_utts;
    // a node holds at most one token per frame; multipath frames also create tokens behind every root
    b->sw.cap = b->lex->multipath ? b->w.nnode + 2 : b->w.tok_cap + 2;
    void *p = nullptr;
    for (int i = 0; i < 2; i++) {
      JAMD_HIP(hipMalloc(&p, U * b->sw.cap * sizeof(STok))); b->owned.push_back(p); b->sw.tl[i] = (STok *)p;
      JAMD_HIP(hipMalloc(&p, U * b->sw.cap * sizeof(int))); b->owned.push_back(p); b->sw.ti[i] = (int *)p;
    }
    JAMD_HIP(hipMalloc(&p, U * b->w.nnode * sizeof(int))); b->owned.push_back(p); b->sw.token = (int *)p;
  }
  b->strict = on != 0;
  if (!on) b->exact = b->exact_status == 0;          // back to the work area's default order (jamd_beam_create())
  return JAMD_OK;
}

int jamd_beam_set_order_mode(jamd_beam *b, int mode) {
  if (!b) { jamd_set_error("jamd_beam_set_order_mode: NULL"); return JAMD_EINVAL; }
  if (b->streaming > 0) { jamd_set_error("jamd_beam_set_order_mode: a streaming session is open"); return JAMD_ESTATE; }
  switch (mode) {
    case JAMD_ORDER_FAST:
      if (b->lex->d.nfwd > 0) { jamd_set_error("jamd_beam_set_order_mode: the canonical-tie kernel does not carry a forward DFA's state"); return JAMD_ESTATE; }
      { const int rc = jamd_beam_set_strict_order(b, 0); b->exact = false; return rc; }
    case JAMD_ORDER_STRICT: b->exact = false; return jamd_beam_set_strict_order(b, 1);
    case JAMD_ORDER_EXACT:
    case JAMD_ORDER_EXACT_SERIAL:
      if (b->exact_status != 0) {
        jamd_set_error("jamd_beam_set_order_mode: the exact-order kernel cannot serve this work area (%s)",
                       b->exact_status == -1 ? "visiting index exceeds 32 bits"
                       : b->exact_status == -2 ? "beam too wide for the LDS image" : b->exact_status == -3 ? "more than 2^21 tokens per frame"
                       : b->exact_status == -4 ? "multipath lexicon in which a root reaches a word end along its own arcs" : "no LDS");
        return JAMD_ESTATE;
      }
      b->xw.prune_mode = b->xw_half.prune_mode = mode == JAMD_ORDER_EXACT_SERIAL ? 1 : 0;
      b->strict = false;
      b->exact = true;
      return JAMD_OK;
    default: jamd_set_error("jamd_beam_set_order_mode: mode=%d", mode); return JAMD_EINVAL;
  }
}

int jamd_beam_set_workgroup_shape(jamd_beam *b, int shape) {
  if (!b) { jamd_set_error("jamd_beam_set_workgroup_shape: NULL"); return JAMD_EINVAL; }
  if (b->streaming > 0) { jamd_set_error("jamd_beam_set_workgroup_shape: a streaming session is open"); return JAMD_ESTATE; }
  if (shape != JAMD_SHAPE_AUTO && shape != JAMD_SHAPE_FULL && shape != JAMD_SHAPE_HALF) {
    jamd_set_error("jamd_beam_set_workgroup_shape: shape=%d", shape); return JAMD_EINVAL;
  }
  if (shape == JAMD_SHAPE_HALF && b->half_status != 0) {
    jamd_set_error("jamd_beam_set_workgroup_shape: beam %d does not fit the half shape (%s)", b->w.beam,
                   b->exact_status != 0 ? "the exact-order kernel cannot serve this work area" : "half a CU's LDS holds no typical frame");
    return JAMD_ESTATE;
  }
  b->shape_mode = shape;
  return JAMD_OK;
}

int jamd_beam_wait_started(jamd_beam *b) {
  if (!b) { jamd_set_error("jamd_beam_wait_started: NULL"); return JAMD_EINVAL; }
  if (!b->ev_started) return JAMD_OK;                  // nothing launched yet
  JAMD_HIP(hipSetDevice(b->eng->device));
  JAMD_HIP(hipEventSynchronize(b->ev_started));
  return JAMD_OK;
}

int jamd_beam_stream_wait_resident(jamd_beam *b, void *stream) {
  if (!b) { jamd_set_error("jamd_beam_stream_wait_resident: NULL"); return JAMD_EINVAL; }
  if (!b->ev_started) return JAMD_OK;                  // nothing launched yet
  JAMD_HIP(hipSetDevice(b->eng->device));
  if (b->d_resident) {
    // the command processor of `stream` waits until the counter the first-pass workgroups bump when they start has
    // reached the latest launch's share; the host is not involved
    JAMD_HIP(hipStreamWaitValue32(jamd_stream(b->eng, stream), b->d_resident, b->resident_target, hipStreamWaitValueGte, 0xffffffffu));
    return JAMD_OK;
  }
  JAMD_HIP(hipEventSynchronize(b->ev_started));       // no wait-on-memory on this device: the host waits, and gives the dispatcher a moment
  struct timespec ms = {0, 1000000};
  nanosleep(&ms, nullptr);
  return JAMD_OK;
}

int jamd_beam_debug_preset_resident(jamd_beam *b, unsigned count) {
  if (!b) { jamd_set_error("jamd_beam_debug_preset_resident: NULL"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(b->eng->device));
  JAMD_HIP(hipDeviceSynchronize());
  if (b->d_resident) JAMD_HIP(hipMemcpy(b->d_resident, &count, sizeof(unsigned), hipMemcpyHostToDevice));
  b->launched_wg = count; b->resident_target = count;
  return JAMD_OK;
}

int jamd_beam_debug_resident(const jamd_beam *b, unsigned *launched, unsigned *target) {
  if (!b) { jamd_set_error("jamd_beam_debug_resident: NULL"); return JAMD_EINVAL; }
  if (launched) *launched = b->launched_wg;
  if (target) *target = b->resident_target;
  return JAMD_OK;
}

int jamd_beam_workgroup_shape(const jamd_beam *b, int nutt) {
  if (!b) return -1;
  return use_half_shape(b, nutt) ? JAMD_SHAPE_HALF : JAMD_SHAPE_FULL;
}

int jamd_beam_exact_layout(const jamd_beam *b) {
  if (!b) return -1;
  return b->exact_status != 0 ? 0 : (b->xw.wide ? 2 : 1);
}

int jamd_beam_order_mode(const jamd_beam *b) {
  if (!b) return -1;
  return b->strict ? JAMD_ORDER_STRICT : b->exact ? (b->xw.prune_mode ? JAMD_ORDER_EXACT_SERIAL : JAMD_ORDER_EXACT) : JAMD_ORDER_FAST;
}

static int prune_order_impl(jamd_beam *b, const float *scores, int n, int *order, int *nkeep, int *arr) {
  if (!b || !scores || !order || !nkeep || n < 1) { jamd_set_error("jamd_beam_prune_order: bad argument"); return JAMD_EINVAL; }
  if (b->exact_status != 0) { jamd_set_error("jamd_beam_prune_order: the exact-order kernel cannot serve this work area"); return JAMD_ESTATE; }
  if (n > (1 << 20)) { jamd_set_error("jamd_beam_prune_order: n=%d too large", n); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(b->eng->device));
  if ((size_t)n > b->pcap) {
    void *p = nullptr;
    const size_t cap = ((size_t)n + 1024 + 3) & ~(size_t)3;     // multiple of 4: the heap and the top-list scratch stay aligned
    JAMD_HIP(hipMalloc(&p, cap * 4)); b->owned.push_back(p); b->d_pkeys = (unsigned *)p;
    // out[cap] + nout (+ pad) | heap u64[cap + 2] | top-list scratch u32x4[beam + 256] (wide layout) | sweep replay scratch
    JAMD_HIP(hipMalloc(&p, 4 * (cap + 16) + 8 * (cap + 2) + 16 * ((size_t)b->w.beam + 256) + xbeam_sweep_bytes(b->w.beam))); b->owned.push_back(p); b->d_pout = (int *)p;
    b->pcap = cap;
    b->d_parr = nullptr;
  }
  if (arr && !b->d_parr) { void *p = nullptr; JAMD_HIP(hipMalloc(&p, 4 * b->pcap)); b->owned.push_back(p); b->d_parr = (int *)p; }
  std::vector<unsigned> keys((size_t)n);
  for (int i = 0; i < n; i++) {
    float f = scores[i] + 0.0f; unsigned u; memcpy(&u, &f, 4);
    keys[i] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
  hipStream_t st = b->eng->stream;
  JAMD_HIP(hipMemcpyAsync(b->d_pkeys, keys.data(), 4 * (size_t)n, hipMemcpyHostToDevice, st));
  int *d_nout = b->d_pout + b->pcap;
  unsigned long long *d_heap = reinterpret_cast<unsigned long long *>(b->d_pout + b->pcap + 16);
  u32x4 *d_collect = reinterpret_cast<u32x4 *>(d_heap + b->pcap + 2);
  unsigned char *d_sweep = reinterpret_cast<unsigned char *>(d_collect + (size_t)b->w.beam + 256);
  xbeam_prune_order_launch((!arr && use_half_shape(b, 1)) ? b->xw_half : b->xw, b->d_pkeys, n, b->w.beam, b->d_pout, d_nout, d_heap, d_collect, d_sweep, arr ? b->d_parr : nullptr, st);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { jamd_set_error("jamd_beam_prune_order: launch failed: %s", hipGetErrorString(le)); return JAMD_ELAUNCH; }
  JAMD_HIP(hipMemcpyAsync(nkeep, d_nout, 4, hipMemcpyDeviceToHost, st));
  JAMD_HIP(hipStreamSynchronize(st));
  if (*nkeep < 0 || *nkeep > n) { jamd_set_error("jamd_beam_prune_order: the kernel reported %d of %d tokens kept", *nkeep, n); return JAMD_ELAUNCH; }
  JAMD_HIP(hipMemcpy(order, b->d_pout, 4 * (size_t)*nkeep, hipMemcpyDeviceToHost));
  if (arr) JAMD_HIP(hipMemcpy(arr, b->d_parr, 4 * (size_t)n, hipMemcpyDeviceToHost));
  return JAMD_OK;
}

int jamd_beam_prune_order(jamd_beam *b, const float *scores, int n, int *order, int *nkeep) {
  return prune_order_impl(b, scores, n, order, nkeep, nullptr);
}

int jamd_beam_prune_arrange(jamd_beam *b, const float *scores, int n, int *order, int *nkeep, int *tindex) {
  if (!tindex) { jamd_set_error("jamd_beam_prune_arrange: bad argument"); return JAMD_EINVAL; }
  return prune_order_impl(b, scores, n, order, nkeep, tindex);
}

int jamd_beam_prune_stats(jamd_beam *b, int utt, int stats[16], int reset) {
  if (!b || !stats || utt < 0 || utt >= b->max_utts) { jamd_set_error("jamd_beam_prune_stats: bad argument"); return JAMD_EINVAL; }
  if (b->exact_status != 0) { jamd_set_error("jamd_beam_prune_stats: the exact-order kernel does not serve this work area"); return JAMD_ESTATE; }
  JAMD_HIP(hipSetDevice(b->eng->device));
  unsigned char *p = b->w.slices + (size_t)utt * b->w.utt_stride + b->xw.o_pstat;
  JAMD_HIP(hipMemcpy(stats, p, 16 * sizeof(int), hipMemcpyDeviceToHost));
  if (reset) JAMD_HIP(hipMemset(p, 0, 16 * sizeof(int)));
  return JAMD_OK;
}

int jamd_beam_prune_info(jamd_beam *b, int *sweep_rounds, int *sweep_us, int *sweep_events) {
  if (!b || !sweep_rounds || !b->d_pout) { jamd_set_error("jamd_beam_prune_info: bad argument (or no jamd_beam_prune_order() call yet)"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(b->eng->device));
  int v[15];
  JAMD_HIP(hipMemcpy(v, b->d_pout + b->pcap + 1, sizeof(v), hipMemcpyDeviceToHost));
  if (getenv("JAMD_SWEEP_PROF")) fprintf(stderr, "sweep phases (us): setup %d  tables %d  level0 %d  levels %d  chains %d  rebuild %d | level phase 1 %d  phase 2 %d\n",
                                         v[3] / 100, v[4] / 100, v[5] / 100, v[6] / 100, v[7] / 100, v[8] / 100, v[9] / 100, v[10] / 100);
  if (getenv("JAMD_SWEEP_PROF") && (v[11] | v[12] | v[13] | v[14])) fprintf(stderr, "sift replay (us): load %d  first window's dependencies %d  sifts %d  output %d\n", v[11] / 100, v[12] / 100, v[13] / 100, v[14] / 100);
  *sweep_rounds = v[0];
  if (sweep_us) *sweep_us = v[1] / 100;            // wall_clock64(): 100 MHz
  if (sweep_events) *sweep_events = v[2];
  return JAMD_OK;
}

int jamd_beam_results(jamd_beam *b, jamd_pass1_result *out, int nutt) {
  if (!b || !out || nutt < 0 || nutt > b->max_utts) { jamd_set_error("jamd_beam_results: bad argument"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(b->eng->device));
  JAMD_HIP(hipDeviceSynchronize());
  if (nutt) JAMD_HIP(hipMemcpy(out, b->w.res, sizeof(jamd_pass1_result) * nutt, hipMemcpyDeviceToHost));
  return JAMD_OK;
}

int jamd_beam_trellis(jamd_beam *b, int utt, jamd_trellis_atom *atoms, int cap, int *natom) {
  if (!b || !natom || utt < 0 || utt >= b->max_utts) { jamd_set_error("jamd_beam_trellis: bad argument"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(b->eng->device));
  JAMD_HIP(hipDeviceSynchronize());
  jamd_pass1_result r;
  JAMD_HIP(hipMemcpy(&r, b->w.res + utt, sizeof(r), hipMemcpyDeviceToHost));
  *natom = r.natom;
  if (atoms) {
    const int n = r.natom < cap ? r.natom : cap;
    if (n > 0) JAMD_HIP(hipMemcpy(atoms, b->w.slices + (size_t)utt * b->w.utt_stride + b->w.o_atoms, sizeof(jamd_trellis_atom) * n,
                                  hipMemcpyDeviceToHost));
  }
  return JAMD_OK;
}

}  // extern "C"
