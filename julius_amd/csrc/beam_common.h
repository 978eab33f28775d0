// beam_common.h -- declarations shared by the first-pass kernels (beam.hip: frame-parallel and strict-order
// kernels, host side; beam_exact.hip: the exact-order frame-parallel kernel).  Everything lives in an
// named namespace (the structs cross translation units), the functions are inline.
#pragma once
#include "jamd_device.h"
#include <type_traits>

namespace jamdb {
using namespace jamd;

constexpr int NT = 1024;                // threads per utterance workgroup
constexpr int MAXSEQ = 150;             // MAXSEQNUM, libsent/include/sent/speech.h:50
constexpr int kMaxDynLds = 159 * 1024;  // dynamic LDS budget of the one workgroup a CU holds (160 KB LDS per CU, < 1 KB static)
constexpr int kHalfNT = 512;              // the exact-order kernel's half shape: two workgroups per CU (beam_exact.h)
constexpr int kHalfDynLds = 79 * 1024;
constexpr int kHistBytes = 2048 * 4;    // rank-select histogram; shares its space with the first cells (free during step D)

// All lexicon / LM arrays live in ONE device allocation and are addressed as base + 32-bit byte
// offset: a kernel that carries some forty 64-bit array pointers next to its per-utterance work
// pointers needs twice the scalar registers the hardware has, and the spilled ones come back
// through v_readlane -- a third of the first-pass kernel's instructions before this layout.  With a
// uniform base the loads also take the `global_load v, voffset32, s[base]` form (no 64-bit address
// arithmetic per access).
#define JAMD_LEX_ARRAYS(X)                                                                       \
  X(int, ac_to) X(float, ac_a)                                                                               \
  X(int2, iso_root)        /* [isolatenum] {root node, successor word scword[scid[root]]}                  */ \
  X(float2, shared_root)   /* [nshared]    {root node bits, fscore[-scid[root]]}                           */ \
  X(int, word_end)         /* [nword] node whose stend is the word                                         */ \
  X(int, startnode) X(int, start2isolate)   /* [startnum] as in wchmm (the strict-order kernel walks them like beam.c) */ \
  X(int, lc_tab) X(int, word_lc) X(int, set_off) X(int, set_states)                                          \
  X(float, wordend_a) X(int, wton) X(float, cprob) X(unsigned char, is_transparent)                          \
  X(int, word_head) X(float, fscore) X(int, scword)                                                          \
  X(float, ng_uni_prob) X(float, ng_uni_bo) X(int, ng_bi_bgn) X(int, ng_bi_num) X(int, ng_bi_wid) X(float, ng_bi_prob) \
  /* grammar (per-category trees): category-pair matrix [ncat][ncat] (dfa_cp()), each root's category         \
     wton[start2wid[root]], the initial tokens [ninit] */                                                      \
  X(unsigned char, cat_pair) X(int, root_cat) X(int, init_node) X(float, init_lscore)                         \
  /* forward DFA (nfwd > 0; libjulius/src/beam.c:1739-1747, :2412-2422): arcs of a state in list order, the initial tokens' states */ \
  X(int, fwd_off) X(int, fwd_label) X(int, fwd_to) X(int, init_to_state)

struct LexDev {
  int nnode, nword, startnum, isolatenum, nshared, nlc, cdset_method, cdmax_num;
  int head_silwid, tail_silwid, ng_mode, ng_unk_id;
  float ng_unk_num_log, lm_weight, lm_penalty, lm_penalty_trans;
  int lm_type, ncat, ninit; float penalty1;
  int nfwd;                           // states of the forward DFA, 0 = none (grammars only)
  const unsigned char *base;          // the arena
  // Cross-word LM table (N-gram lexicons): iwtab[ctx * isolatenum + i] = bigram_prob(ctx, wton(w_i)) + cprob(w_i) for the
  // word w_i behind isolated root i -- every entry max_successor_prob_iw() (factoring_sub.c:1049-1143) can ever put
  // into its per-last-word cache iw_sc_cache, computed once when the lexicon is created (same arithmetic, same
  // floats) and kept in HBM: 20 000 contexts x 189 roots = 15 MB.  NULL when it would exceed kIwTabMaxBytes.
  const float *iwtab;
#define X(T, name) unsigned o_##name;
  JAMD_LEX_ARRAYS(X)
#undef X
  template <typename T>
  __device__ __forceinline__ T at(unsigned off, int i) const {
    return *reinterpret_cast<const T *>(base + (unsigned)(off + (unsigned)i * (unsigned)sizeof(T)));
  }
#define X(T, name)                                                                   \
  __device__ __forceinline__ T name(int i) const { return at<T>(o_##name, i); }      \
  __device__ __forceinline__ const T *name##_ptr() const { return reinterpret_cast<const T *>(base + o_##name); }
  JAMD_LEX_ARRAYS(X)
#undef X
  // A node is ONE 32-byte record (round 6): {self_a bits, next_a bits, ac_off, ac_end} (wchmm->self_a / next_a / ac) and
  // {stend, scid, out_id, out_kind} (stend, state[].scid, outstyle) -- a survivor's transitions, the successor id of its
  // next node (the following record) and the record of the token it creates there come out of one or two 64-byte sectors,
  // where three arrays took three or four.
  unsigned o_node_a, o_node_b, o_scid;      // the record's first half, its second half (+ 16), scid inside it (+ 20)
  __device__ __forceinline__ int4 node_a(int i) const { return *reinterpret_cast<const int4 *>(base + (unsigned)(o_node_a + 32u * (unsigned)i)); }
  __device__ __forceinline__ int4 node_b(int i) const { return *reinterpret_cast<const int4 *>(base + (unsigned)(o_node_b + 32u * (unsigned)i)); }
  __device__ __forceinline__ int scid(int i) const { return *reinterpret_cast<const int *>(base + (unsigned)(o_scid + 32u * (unsigned)i)); }
};

struct __attribute__((aligned(16))) Tok {   // TOKEN2, libjulius/include/julius/beam.h:35-45
  int node; float score; int last_tre; int last_cword;
  float last_lscore; int last_wid; int pad0, pad1;   // last_wid = wid of atoms[last_tre] (-1 for bos)
};

struct StreamState {     // what a streaming utterance carries from one launch to the next
  int started, active, frames_done, n_surv, n_atom, ties, ties_we, ties_cut, max_tokens;
  float thr;
};

struct Work {            // per-utterance slices are addressed with the strides below
  StreamState *stream;           // [utt] (allocated by jamd_beam_stream_begin)
  unsigned *resident;            // workgroups of the first-pass kernels that have started, ever (signal memory: a stream can wait
                                 // on it, jamd_beam_stream_wait_resident()), or nullptr
  // Per-utterance arrays live in ONE slice per utterance (slices + utt * utt_stride) and are
  // addressed as slice base + 32-bit offset, for the same reason as the lexicon arena (LexDev):
  unsigned char *slices; unsigned long long utt_stride;
  unsigned o_nodekey;            // u64  [nnode]    Viterbi cells (0 = empty)
  unsigned o_cur;                // Tok  [tok_cap]  tokens created this frame
  unsigned o_cur_key;            // u32  [tok_cap]  their order-preserving score bits (compact, for the rank select)
  unsigned o_touched;            // int2 [tok_cap]  nodes touched this frame: {node, LDS cell slot or -1 = nodekey[]}
  unsigned o_arcq;               // int2 [tok_cap]  work queue of (survivor, extra arc) pairs
  unsigned o_atoms;              // jamd_trellis_atom [atom_cap]
  unsigned o_lmcache;            // u64  [nscword]  LM memo, see below
  unsigned o_sv;                 // survivor image (sv_bytes) when it does not live in LDS / between streaming launches
  jamd_pass1_result *res;        // [utt]
  // the survivor state lives in LDS when it fits (sv_bytes of dynamic shared memory), else at o_sv;
  // the LM memo holds (context N-gram id << 32 | prob bits) per successor id: the reference's
  // LM_PROB_CACHE (wchmm.h:117-149, factoring_sub.c:965-986)
  int nscword;
  int sv_bytes, use_lds, hsize;  // hsize = slots of the node -> survivor hash (power of two)
  int cell_slots;                // LDS Viterbi cells of the current frame (power of two, 0 = all cells in nodekey[])
  int lds_bytes;                 // dynamic LDS per workgroup without the score-row cache
  int cell_off, node_off, row_off;  // byte offsets in dynamic LDS: cells / histogram, cell owners, score row
  int row_cache;                 // set per launch: the frame's [nstate] score row is copied to LDS
  int tok_cap, atom_cap, beam, nnode, nword;
  float width;
};

// order-preserving map float -> u32 (larger float <=> larger unsigned)
__device__ __forceinline__ unsigned ord(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unord(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// The arc of forward-DFA state `st` labelled `cat` (first match in list order, libjulius/src/beam.c:1741-1746, :2415-2420):
// its target state, or -1.  A token whose own state is -1 (no arc matched when it was created) finds nothing: the reference
// would index st[-1] there -- such a token cannot arise from a consistent pair of automata.
__device__ __forceinline__ int fwd_next(const LexDev &lx, int st, int cat) {
  if (st < 0 || st >= lx.nfwd) return -1;
  const int a0 = lx.fwd_off(st), a1 = lx.fwd_off(st + 1);
  for (int a = a0; a < a1; a++) if (lx.fwd_label(a) == cat) return lx.fwd_to(a);
  return -1;
}

// search_bigram(), ngram_access.c:225-247
__device__ __forceinline__ int search_bigram(const LexDev &lx, int w_context, int w) {
  int left = lx.ng_bi_bgn(w_context);
  if (left < 0) return -1;
  int right = left + lx.ng_bi_num(w_context) - 1;
  while (left < right) {
    const int mid = (left + right) / 2;
    if (lx.ng_bi_wid(mid) < w) left = mid + 1; else right = mid;
  }
  return (lx.ng_bi_wid(left) == w) ? left : -1;
}

// ngram->bigram_prob as chosen by bi_prob_func_set(), ngram_access.c:288-466
__device__ inline float bigram_prob(const LexDev &lx, int w1, int w2) {
  int n2; float prob;
  if (lx.ng_mode == JAMD_NG_NORMAL || lx.ng_mode == JAMD_NG_ADDITIONAL_OLD) {
    if ((n2 = search_bigram(lx, w1, w2)) >= 0) prob = lx.ng_bi_prob(n2);
    else prob = lx.ng_uni_bo(w1) + lx.ng_uni_prob(w2);
  } else if (lx.ng_mode == JAMD_NG_ADDITIONAL) {
    if ((n2 = search_bigram(lx, w2, w1)) >= 0) prob = lx.ng_bi_prob(n2);
    else prob = lx.ng_uni_bo(w1) + lx.ng_uni_prob(w2);
  } else {
    if ((n2 = search_bigram(lx, w2, w1)) >= 0) prob = lx.ng_bi_prob(n2);
    else prob = lx.ng_uni_bo(w2) + lx.ng_uni_prob(w1);
    prob = prob + lx.ng_uni_prob(w2) - lx.ng_uni_prob(w1);
  }
  if (w2 != lx.ng_unk_id) return prob;
  return prob - lx.ng_unk_num_log;
}

// max_successor_prob(), factoring_sub.c:942-1008 (UNIGRAM_FACTORING), scid given.
// `memo` is the per-utterance equivalent of the reference's lastwcache/probcache pair:
// one (context, value) entry per successor id, a pure memo of the 2-gram lookup.  A token
// waiting in front of a branch asks for the same pair every frame, so nearly every call
// is one 8-byte load instead of a binary search.  Entries are written as single 64-bit
// words, so concurrent writers cannot tear them; NULL disables the memo.
__device__ __forceinline__ float max_successor_prob(const LexDev &lx, int lastword, int scid,
                                                    unsigned long long *memo = nullptr) {
  if (lastword < 0) return 0.0f;
  if (scid < 0) return lx.fscore(-scid);
  const int ctx = lx.wton(lastword);
  if (memo) {
    const unsigned long long m = memo[scid];
    if ((int)(unsigned)(m >> 32) == ctx) return __uint_as_float((unsigned)m);
  }
  const int w = lx.scword(scid);
  const float p = bigram_prob(lx, ctx, lx.wton(w)) + lx.cprob(w);
  if (memo) memo[scid] = ((unsigned long long)(unsigned)ctx << 32) | __float_as_uint(p);
  return p;
}

// outprob_style(), outprob_style.c:354-486, with the name lookups replaced by
// the flattened left-context table
__device__ inline float node_outprob(const LexDev &lx, const float *__restrict__ row, int kind, int id, int last_wid) {
  int ent;
  if (kind == JAMD_AS_STATE) return row[id];
  if (kind == JAMD_AS_LSET) ent = ~id;
  else ent = lx.lc_tab((size_t)id * (lx.nlc + 1) + (last_wid < 0 ? lx.nlc : lx.word_lc(last_wid)));
  if (ent >= 0) return row[ent];
  ent = ~ent;
  return cd_reduce(row, lx.set_states_ptr(), lx.set_off(ent), lx.set_off(ent + 1), lx.cdset_method, lx.cdmax_num);
}

// the state (>= 0) or ~state-set (< 0) that outprob_style() scores for a node
__device__ __forceinline__ int outprob_entry(const LexDev &lx, int kind, int id, int last_wid) {
  if (kind == JAMD_AS_STATE) return id;
  if (kind == JAMD_AS_LSET) return ~id;
  return lx.lc_tab((size_t)id * (lx.nlc + 1) + (last_wid < 0 ? lx.nlc : lx.word_lc(last_wid)));
}

// the acoustic scores of the frame being finalized: the [nstate] row in global memory, or its copy
// in LDS (a frame makes some 15 000 gathers from it: one per new token plus the members of every
// state set -- a third of all the divergent loads of the frame)
struct RowRef {
  const float *g; const float *l; bool lds;
  __device__ __forceinline__ float operator[](int i) const { return lds ? l[i] : g[i]; }
};

struct Shared {
  unsigned long long we_best;       // (ord(score + wordend_a), word that ended)
  int n_new, n_we, n_arc, n_atom, n_surv, ties, ties_we, ties_cut, best_atom;
  unsigned maxbits, minbits;
  unsigned sel_digit, sel_need, sel_count;
  unsigned wsum[NT / 64];            // per-wave histogram totals of the rank select
  int eq_n, eq_node[128];           // tokens exactly on the rank cut (tie handling)
};

// LDS arrays are addressed through pointers that CARRY the address space: a generic pointer that the
// compiler cannot trace back to LDS (through a struct, a select, a non-inlined call) becomes flat_load /
// flat_store -- two to three times the latency of ds_read / ds_write and no loop unrolling
// (measured: the rank-by-counting loop below ran 14x slower through a generic pointer).
#define JAMD_LDS __attribute__((address_space(3)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef JAMD_LDS unsigned long long lds_u64;
typedef JAMD_LDS unsigned lds_u32;
typedef JAMD_LDS int lds_i32;
typedef JAMD_LDS float lds_f32;
typedef JAMD_LDS u32x4 lds_v4;

// a token record (two 16-byte quads) to / from LDS
__device__ __forceinline__ Tok lds_tok_load(const lds_v4 *p, int j) {
  const u32x4 a = p[2 * j], b = p[2 * j + 1];
  Tok t;
  t.node = (int)a.x; t.score = __uint_as_float(a.y); t.last_tre = (int)a.z; t.last_cword = (int)a.w;
  t.last_lscore = __uint_as_float(b.x); t.last_wid = (int)b.y; t.pad0 = (int)b.z; t.pad1 = (int)b.w;
  return t;
}
__device__ __forceinline__ void lds_tok_store(lds_v4 *p, int j, const Tok &t) {
  u32x4 a, b;
  a.x = (unsigned)t.node; a.y = __float_as_uint(t.score); a.z = (unsigned)t.last_tre; a.w = (unsigned)t.last_cword;
  b.x = __float_as_uint(t.last_lscore); b.y = (unsigned)t.last_wid; b.z = (unsigned)t.pad0; b.w = (unsigned)t.pad1;
  p[2 * j] = a; p[2 * j + 1] = b;
}

// The survivor image of the frame-parallel kernel (tokens, the atom each word end emitted, the frame's word-end
// list, node -> survivor hash), in LDS (typed pointers, ds_* instructions) or -- beams too wide for LDS -- in the
// utterance's global slice (plain pointers).  One kernel instantiation per case: a run-time select between the two
// bases turns every access into a flat_* instruction at two to three times the LDS latency.
template <bool LDS> struct SvImage;
template <> struct SvImage<true> {
  lds_v4 *tok; lds_i32 *atom, *we, *hkey, *hval;
  __device__ __forceinline__ void bind(unsigned char *lds_base, unsigned char *, int beam, int hsize) {
    tok = (lds_v4 *)lds_base;
    atom = (lds_i32 *)(lds_base + (size_t)beam * sizeof(Tok));
    we = atom + beam; hkey = we + beam; hval = hkey + hsize;
  }
  __device__ __forceinline__ Tok load(int j) const { return lds_tok_load(tok, j); }
  __device__ __forceinline__ void store(int j, const Tok &t) const { lds_tok_store(tok, j, t); }
};
template <> struct SvImage<false> {
  Tok *tok; int *atom, *we, *hkey, *hval;
  __device__ __forceinline__ void bind(unsigned char *, unsigned char *glob_base, int beam, int hsize) {
    tok = (Tok *)glob_base;
    atom = (int *)(glob_base + (size_t)beam * sizeof(Tok));
    we = atom + beam; hkey = we + beam; hval = hkey + hsize;
  }
  __device__ __forceinline__ Tok load(int j) const { return tok[j]; }
  __device__ __forceinline__ void store(int j, const Tok &t) const { tok[j] = t; }
};

// node -> survivor index, open addressing (survivor nodes are distinct)
__device__ __forceinline__ unsigned hslot(int node, int hmask) {
  return ((unsigned)node * 2654435761u >> 7) & (unsigned)hmask;
}
template <typename KP, typename VP>
__device__ __forceinline__ void hash_put(KP hkey, VP hval, int hmask, int node, int j) {
  unsigned h = hslot(node, hmask);
  while (atomicCAS((int *)&hkey[h], -1, node) != -1) h = (h + 1) & (unsigned)hmask;
  hval[h] = j;
}
template <typename KP, typename VP>
__device__ __forceinline__ int hash_get(KP hkey, VP hval, int hmask, int node) {
  unsigned h = hslot(node, hmask);
  for (int guard = 0; guard <= hmask; guard++) {
    const int k = hkey[h];
    if (k == node) return hval[h];
    if (k == -1) break;
    h = (h + 1) & (unsigned)hmask;
  }
  return 0;   // unreachable for a live source
}

// Wave-aggregated slot allocation: the active lanes that want a slot are counted with a
// ballot, ONE lane bumps the shared counter, every lane takes base + its rank.  Cuts the
// same-address LDS atomics (thousands per frame on n_new / n_atom / n_surv) by up to 64x.
__device__ __forceinline__ int wave_alloc(int *counter, bool want) {
  const unsigned long long m = __ballot(want);
  if (!want) return -1;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, leader, 64);
  return base + __popcll(m & ((1ull << lane) - 1ull));
}

}  // namespace jamdb
