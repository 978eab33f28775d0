// beam_exact_mp.h -- multipath lexicons (hmminfo->multipath) on the exact-order frame-parallel machinery.
// Included by beam_exact.hip inside its anonymous namespace, behind the kernel for ordinary lexicons: the helpers
// (cells, creation-order bitmap, heaps, exact_prune()) and the slice macros are that file's.
//
// The reference runs a different frame for such models (beam.c:2747-2836, :2930-2943, :3066-3073; the frame as
// decoded one lane per utterance by beam_strict_mp_kernel, beam.hip):
//   1  word-internal transitions of every survivor                               (steps 0', A', C1: tokens WITHOUT this
//                                                                                  frame's output probability)
//   2  sort_token_no_order() over the NEW tokens                                  (step M)
//   3  word ends among the beam's best of them, in tindex order: trellis word, cross-word transition -- the root has no
//      output, so the token goes on along the root's own arcs inside the frame (:2467-2510); a target that already
//      holds a token is improved in place, a new one is appended                  (steps B0, B', C2)
//   4  output probabilities on emitting nodes only                                (step O)
//   5  sort_token_no_order() over all tokens -- starting from tindex[] AS STEP 2 LEFT IT (residual heap + extracted
//      part) plus the appended tokens                                             (step D)
// Frame 0 goes through this too (the initial tokens carry the LM score only, :1657, :1733), and one transition-only
// call ends the input (t == T: steps 1-3 without the cross-word part).
//
// Step 5's input is the whole array of step 2 -- residual heap included.  exact_prune<FULL> delivers it: the extracted
// part is the sweep replay's extraction order, the residual heap comes out of the sift replay below the extracted region
// (beam_sweep.h: written for the downward sort, mirrored for the upward one); where that machinery cannot run (heap
// outside LDS, too many events) step 2 is carried out literally (heapify + the extraction loop, pipelined on one wave).
// Step 5 itself is exact_prune() over the keys gathered in that arrangement.
// (A partial heap sort over DISTINCT scores extracts in score order whatever the arrangement, so tindex[] matters to
// step 5 only when two of its survivors -- or the last survivor and the first loser -- tie, or when it runs downward.
// Measured on the 20 000-word task: that is 99 % of the frames -- pronunciation variants and homophones tie exactly --,
// so a variant that sorted literally only on demand was slower; profiles/r04_multipath_*.)
//
// Visiting indices.  First half: (survivor position << s1) | transition number, as in the ordinary kernel.  Second
// half: (rank of the word end among the frame's word ends << s1) | (root number * XW + transition number of the root);
// the factoring pass (beam_inter_word_factoring(), one source: the best word end) uses rank = number of word ends.
// A node that holds a token of the first half is found through nodetok[] (token id + 1, cleared in step O) -- one entry
// per node A ROOT LEADS TO (jamd_lexicon::o_mp_tgt numbers them; no other node meets a token of the second half): a few
// thousand words that stay in L2, where a table over all nodes was a megabyte per utterance written four bytes at a time
// (half of the frame's memory-side traffic, profiles/traffic_first_pass.json).
//
// One restriction, checked when the lexicon is loaded (jamd_lexicon::mp_parallel): no root may reach a word-end node
// along its own arcs (a word made of tee models only).  There a cross-word transition would improve a word end that the
// loop of step 3 has yet to visit -- or has visited already -- and the outcome depends on the loop's position; such
// lexicons stay on the strict-order kernel.  (The reference builds none: wchmm_add_word() refuses a word whose models can
// all be skipped, wchmm.c:1345-1362 -- only a hand-written descriptor gets here.)

// sort_token_no_order() (beam.c:1492 over :1342-1480) carried out literally on (score bits << 32 | index) entries:
// H[1..n] ends as tindex[0..n-1].  The heap in LDS when it fits (pipelined extraction), else in the slice (one lane).
template <int NT>
__device__ __forceinline__ void literal_sort(const unsigned *keys, int n, int k, lds_u64 *Hl, int heap_cap, unsigned long long *Hg) {
  const int tid = tid_now();
  const bool upward = k < n - k;
  auto run = [&](auto Hh) -> void {
    constexpr bool kLds = std::is_same<decltype(Hh), lds_u64 *>::value;
    for (int i = tid; i < n; i += NT) Hh[i + 1] = ((unsigned long long)keys[i] << 32) | (unsigned)i;
    if (tid == 0) Hh[0] = 0ull;
    __syncthreads();
    bool heaped = false;
    if constexpr (kLds) heaped = upward ? heapify_overlapped<true, NT>(Hh, n) : heapify_overlapped<false, NT>(Hh, n);
    if (!heaped) { if (upward) heapify_levels<true, NT>(Hh, n); else heapify_levels<false, NT>(Hh, n); }
    if constexpr (kLds) {
      if (tid < 64) { if (upward) heap_extract_pipelined<true>(Hh, n, k); else heap_extract_pipelined<false>(Hh, n, n - k); }
    } else {
      if (tid == 0) { if (upward) heap_extract_serial<true>(Hh, n, k); else heap_extract_serial<false>(Hh, n, n - k); }
    }
    __syncthreads();
  };
  if (n <= heap_cap) run(Hl); else run(Hg);
}

// the multipath kernel's per-frame view of the launch constants (see xargs_now() in beam_exact.hip)
#define XBEAM_MP_VIEWS(KA)                                                                                           \
  const LexDev &lx = (KA).lx; const XWork &xw = (KA).xw; const Work &wk = xw.w;                                        \
  XSv<WIDE> sv;                                                                                                       \
  if constexpr (WIDE) sv.p = reinterpret_cast<u32x4 *>(ub + wk.o_sv); else sv.p = (lds_v4 *)dyn_lds;                    \
  lds_i32 *welist = (lds_i32 *)(dyn_lds + xw.off_we);      /* token ids of the frame's word ends; the final cut returns its order here */ \
  lds_i32 *dbase = (lds_i32 *)(dyn_lds + xw.off_dbase);                                                                \
  lds_u32 *tpre = (lds_u32 *)(dyn_lds + xw.off_tpre);                                                                  \
  XCells cl;                                                                                                          \
  cl.ub = ub; cl.o_nodekey = wk.o_nodekey; cl.o_nodefirst = xw.o_nodefirst; cl.o_touched = wk.o_touched;               \
  cl.nslot = xw.nslot;                                                                                                \
  cl.lkey = (lds_u64 *)(dyn_lds + xw.off_cells);                                                                       \
  cl.lnode = (lds_i32 *)(dyn_lds + xw.off_lnode);                                                                      \
  cl.lfirst = (lds_u32 *)(dyn_lds + xw.off_lfirst);                                                                    \
  lds_f32 *rowc = (lds_f32 *)(dyn_lds + xw.off_row);                                                                   \
  PruneMem pm;                                                                                                        \
  pm.compR = (lds_u64 *)(dyn_lds + xw.off_compr); pm.compT = pm.compR + xw.b_cap;                                      \
  pm.vposR = (lds_u32 *)(dyn_lds + xw.off_vpos);                                                                       \
  pm.idR = (lds_u32 *)(dyn_lds + xw.off_id);                                                                           \
  pm.idT = (lds_u32 *)(dyn_lds + xw.off_idt);                                                                          \
  pm.hist = (lds_u32 *)(dyn_lds + xw.off_hist);                                                                        \
  pm.tailmask = (lds_u32 *)(dyn_lds + xw.off_tail);                                                                    \
  pm.cand = (lds_i32 *)(pm.tailmask + (xw.w.beam + 31) / 32 + 2);                                                      \
  pm.occ = pm.cand + kMaxCand; pm.need = pm.occ + kMaxCand; pm.takers = pm.need + kMaxCand + 4;                        \
  pm.ordv = pm.takers + (kMaxCand + 1) * (kTakers + 1);                                                                \
  pm.b_cap = xw.b_cap;                                                                                                \
  pm.sw_region = (unsigned char JAMD_LDS *)(dyn_lds + xw.off_dov); pm.sw_bytes = xw.off_row - xw.off_dov;             \
  pm.sw_glob = xw.o_sweep ? ub + xw.o_sweep : nullptr;                                                                 \
  pm.pstat = xw.o_sweep ? sh.pst : nullptr;                                                                           \
  lds_u64 *Hlds = (lds_u64 *)(dyn_lds + xw.off_heap);                                                                  \
  unsigned long long *Hglob = reinterpret_cast<unsigned long long *>(ub + xw.o_heap);                                  \
  u32x4 *Gcol = reinterpret_cast<u32x4 *>(ub + xw.o_collect);                                                          \
  auto clear_cells = [&]() { for (int i = tid; i < cl.nslot; i += NT) { cl.lkey[i] = 0ull; cl.lnode[i] = -1; cl.lfirst[i] = 0u; } }; \
  const float lmw = lx.lm_weight, pen = lx.lm_penalty;                                                                 \
  const int lmt = lx.lm_type & 0xff;                                                                                  \
  const bool dfa = lmt != JAMD_LM_NGRAM;                                                                               \
  const bool wordmode = lmt == JAMD_LM_WORD;                                                                           \
  unsigned long long *memo = reinterpret_cast<unsigned long long *>(ub + wk.o_lmcache);                                \
  const int s1 = xw.s1, XW = xw.xw;                                                                                    \
  const unsigned submask = (1u << s1) - 1u;                                                                            \
  const int nroot_x = wordmode ? 0 : (dfa ? lx.startnum : lx.isolatenum);                                              \
  const int slots2 = nroot_x * XW;                         /* visiting indices a word end owns in the second half */ \
  lds_u32 *bm_l = (lds_u32 *)(dyn_lds + xw.off_bm);                                                                    \
  unsigned *bm_g = reinterpret_cast<unsigned *>(ub + xw.o_bitmap);                                                     \
  (void)welist; (void)dbase; (void)tpre; (void)rowc; (void)Hlds; (void)Hglob; (void)Gcol; (void)lmw; (void)pen; (void)memo; \
  (void)XW; (void)submask; (void)slots2; (void)bm_l; (void)bm_g; (void)clear_cells

template <bool TIMED, bool WIDE, int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4)))
beam_exact_mp_kernel(XKArgs ka_, const float *__restrict__ scores, int S, const int *__restrict__ utt_off, int smode) {
  __shared__ XShared sh;
  extern __shared__ __align__(16) unsigned char dyn_lds[];
#if JAMD_XARGS_RELOAD
  const XKArgs &ka0 = xargs_now();
#else
  const XKArgs &ka0 = ka_;
#endif
  if (threadIdx.x == 0 && ka0.xw.w.resident) __hip_atomic_fetch_add(ka0.xw.w.resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int u = min(max(utt_off[gridDim.x + 1 + blockIdx.x], 0), (int)gridDim.x - 1);
  int tid = threadIdx.x;
  const int t_begin = utt_off[u], nrows = utt_off[u + 1] - t_begin;
  StreamState *ss = smode ? ka0.xw.w.stream + u : nullptr;
  const bool resume = smode && ss->started;
  const int base = resume ? ss->frames_done : 0;
  const int T = base + nrows;
  const bool finish = smode != 1;
  unsigned char *const ub = ka0.xw.w.slices + (size_t)u * ka0.xw.w.utt_stride;
#define NODETOK(i) SLICE(unsigned, xw.o_nodetok, i)       /* i = TGT(node) >= 0 */
#define TGT(node) lx.at<int>(xw.o_mp_tgt, node)
#define ARR(i) SLICE(int, xw.o_arr, i)
#define KEY2(i) SLICE(unsigned, xw.o_key2, i)
  jamd_pass1_result *res = ka0.xw.w.res + u;
  XBEAM_MP_VIEWS(ka0);
  int *const pstat_glob = xw.o_sweep ? reinterpret_cast<int *>(ub + xw.o_pstat) : nullptr;
  if (tid == 0) for (int i = 0; i < 16; i++) sh.pst[i] = 0;
  const int head_root = dfa ? -1 : lx.word_head(lx.head_silwid);

  if (resume) {
    if (!ss->active) return;
    if constexpr (!WIDE) {
      const u32x4 *src = (const u32x4 *)(ub + wk.o_sv);
      for (int i = tid; i < wk.sv_bytes / 16; i += NT) sv.p[i] = src[i];
    }
    if (tid == 0) { sh.n_atom = ss->n_atom; sh.n_surv = ss->n_surv; }
    clear_cells();
    __syncthreads();
  } else {
    if (tid == 0) {
      sh.n_atom = 0; sh.n_surv = 0;
      res->status = JAMD_PASS1_OK; res->natom = 0; res->wnum = 0; res->score = JAMD_LOG_ZERO;
      res->died_at = -1; res->ties = 0; res->ties_node = 0; res->ties_wordend = 0; res->ties_cut = 0;
      res->frames = T; res->max_tokens = 0;
      for (int i = 0; i < 8; i++) res->phase_us[i] = 0;
    }
    for (int i = tid; i < wk.nscword; i += NT) memo[i] = 0xffffffff00000000ull;
    // (an utterance can end between steps C1 and O -- the transition-only last call, an overflow -- and leave entries behind)
    for (int i = tid; i < xw.n_mp_tgt; i += NT) NODETOK(i) = 0u;
    __syncthreads();
    if (nrows <= 0) {
      if (tid == 0) { if (smode != 1) res->status = JAMD_PASS1_FAIL; if (ss) { ss->started = 0; ss->active = 1; } }
      return;
    }
    // get_back_trellis_init(): the initial tokens carry the LM score only (init_nodescore :1657-1663, :1733-1737), then
    // sort_token_no_order() (:1807)
    if (!dfa) {
      if (tid == 0) {
        const int4 nr = lx.node_b(head_root);
        Tok nw;
        float ls = (nr.y != 0) ? max_successor_prob(lx, -1, nr.y) : 0.0f;
        ls = ls * lmw + pen;
        nw.node = head_root; nw.last_tre = -1; nw.last_cword = -1; nw.last_wid = -1; nw.last_lscore = ls;
        nw.score = ls; nw.pad0 = nr.x; nw.pad1 = 0;
        sv.store(0, nw);
        sh.n_surv = 1;
      }
      clear_cells();
      __syncthreads();
    } else {
      const int ninit = lx.ninit;
      if (tid == 0) { sh.maxbits = ord(JAMD_LOG_ZERO); sh.minbits = 0xffffffffu; }
      __syncthreads();
      unsigned mymax = ord(JAMD_LOG_ZERO), mymin = 0xffffffffu;
      for (int e = tid; e < ninit; e += NT) {
        const int node = lx.init_node(e);
        Tok nw;
        nw.node = node; nw.last_tre = -1; nw.last_cword = -1; nw.last_wid = -1; nw.last_lscore = lx.init_lscore(e);
        nw.score = nw.last_lscore; nw.pad0 = lx.node_b(node).x; nw.pad1 = lx.nfwd ? lx.init_to_state(e) : 0;   // forward-DFA state (:1739-1747)
        CUR(e) = nw;
        const unsigned b = ordz(nw.score);
        CURKEY(e) = b;
        mymax = max(mymax, b); mymin = min(mymin, b);
      }
      atomicMax(&sh.maxbits, mymax); atomicMin(&sh.minbits, mymin);
      __syncthreads();
      // (sort_token_no_order() :1807; more initial tokens than the beam holds: literally, as in step M)
      int n_keep = ninit, lo = 0;
      if (ninit > wk.beam) {
        const int k = wk.beam;
        literal_sort<NT>(&CURKEY(0), ninit, k, Hlds, xw.heap_cap, Hglob);
        n_keep = k; lo = (k < ninit - k) ? ninit - k : 0;
        for (int j = tid; j < n_keep; j += NT)
          welist[j] = (int)(unsigned)(ninit <= xw.heap_cap ? Hlds[lo + j + 1] : Hglob[lo + j + 1]);
      } else {
        for (int j = tid; j < n_keep; j += NT) welist[j] = j;
      }
      __syncthreads();
      for (int j = tid; j < n_keep; j += NT) sv.store(j, CUR(welist[j]));
      if (tid == 0) sh.n_surv = n_keep;
      clear_cells();
      __syncthreads();
    }
  }
  float thr = resume ? ss->thr : JAMD_LOG_ZERO;
  // the phase clocks of the instrumented instantiation live in LDS (thread 0 adds to them): eight 64-bit counters in
  // registers cost the kernel 16 VGPRs it does not have
  unsigned long long *const ph = sh.ph;
  if (TIMED && threadIdx.x == 0) for (int i = 0; i < 8; i++) sh.ph[i] = 0ull;
  unsigned long long tc = wall_clock64(), tc2 = tc;
  (void)tc2;
  int max_tokens = resume ? ss->max_tokens : 1;
  bool stopped = false;
  __syncthreads();

  auto row_request = [&](int tt) {
    if (!wk.row_cache || tt >= T) return;
    const float *rg = scores + (size_t)(t_begin + tt - base) * S;
    const int ln = tid & 63;
    for (int b = uni((int)(tid >> 6)) * 64; b < S; b += NT)
      if (b + ln < S) __builtin_amdgcn_global_load_lds((glb_void *)(rg + b + ln), (lds_void *)(rowc + b), 4, 0, 0);
  };
  row_request(base);

  for (int t = base; t <= (finish ? T : T - 1); t++) {
    tid = tid_now();
#if JAMD_XARGS_RELOAD
    XBEAM_MP_VIEWS(xargs_now());                           // this frame's view of the launch constants
#endif
    const int n_surv = uni(sh.n_surv);
    __syncthreads();
    if (tid == 0) {
      sh.n_new = 0; sh.n_we = 0; sh.n_arc = 0; sh.we_best = 0ull;
      sh.maxbits = ord(JAMD_LOG_ZERO); sh.minbits = 0xffffffffu; sh.emaxbits = ord(JAMD_LOG_ZERO);
    }
    const bool final_call = (t == T);
    // ---- 0': dense visiting indices of the word-internal transitions (XW per live survivor)
    int nbits = 0;
    {
      int carry = 0;
      for (int j0 = 0; j0 < n_surv; j0 += NT) {
        const int j = j0 + tid;
        int cnt = 0;
        if (j < n_surv) {
          u32x4 a_, b_;
          sv.quads(j, a_, b_);
          const float sc = __uint_as_float(a_.y);
          if (sc > JAMD_LOG_ZERO && !(sc < thr)) cnt = XW;
        }
        const int ex = block_excl_scan<NT>(sh, cnt);
        if (j < n_surv) dbase[j] = carry + ex;
        carry += sh.scan_total;
        __syncthreads();
      }
      if (tid == 0) dbase[n_surv] = carry;
      nbits = carry;
    }
    __syncthreads();
    // ---- A': word-internal transitions (beam_intra_word() :2154-2180)
    auto intra_candidate = [&](const Tok &tk, int j, int next_node, float a, int sub, int nscid) {
      float tmpsum = tk.score + a;
      if (nscid != 0) {
        const float ng = max_successor_prob(lx, tk.last_cword, nscid, memo) * lmw + pen;
        tmpsum -= tk.last_lscore;
        tmpsum += ng;
      }
      xpush(sh, cl, next_node, tmpsum, ((unsigned)j << s1) | (unsigned)sub);
    };
    for (int j = tid; j < n_surv; j += NT) {
      const Tok tk = sv.load(j);
      if (tk.score <= JAMD_LOG_ZERO) continue;
      if (tk.score < thr) continue;
      const int node = tk.node;
      const int4 na = lx.node_a(node);
      const int nscid1 = (!dfa && node + 1 < lx.nnode) ? lx.scid(node + 1) : 0;
      const int e0 = na.z, e1 = na.w;
      if (e1 > e0) {
        const int b0 = atomicAdd(&sh.n_arc, e1 - e0);
        for (int e = e0; e < e1; e++) ARCQ(b0 + e - e0) = make_int2(j | ((2 + e - e0) << 16), e);
      }
      { const float a = __int_as_float(na.x); if (a != JAMD_LOG_ZERO) intra_candidate(tk, j, node, a, 0, 0); }
      { const float a = __int_as_float(na.y); if (a != JAMD_LOG_ZERO) intra_candidate(tk, j, node + 1, a, 1, nscid1); }
    }
    __syncthreads();
    {
      const int n_arc = uni(sh.n_arc);
      for (int q = tid; q < n_arc; q += NT) {
        const int2 it = ARCQ(q);
        const int j = it.x & 0xffff;
        const int to = lx.ac_to(it.y);
        const Tok tk = sv.load(j);
        intra_candidate(tk, j, to, lx.ac_a(it.y), it.x >> 16, (!dfa && to != tk.node) ? lx.scid(to) : 0);
      }
      __syncthreads();
    }
    // ---- C1: the new tokens in creation order, without output probabilities
    const int n1 = uni(sh.n_new);
    if (n1 > wk.tok_cap) {
      if (tid == 0) res->status = JAMD_PASS1_OVERFLOW;
      stopped = true;
      __syncthreads();
      break;
    }
    // creation order = rank of the node's first visit: bitmap over the dense visiting indices + prefix counts
    auto rank_setup = [&](int nb, int ntouch, auto dense_of, auto wants) -> int {   // returns log2(words per thread)
      const int nwords = (nb + 31) >> 5;
      const bool in_lds = nwords <= xw.bm_words;
      for (int i = tid; i < nwords; i += NT) { if (in_lds) bm_l[i] = 0u; else bm_g[i] = 0u; }
      __syncthreads();
      for (int s = tid; s < ntouch; s += NT) {
        const int2 t2 = TOUCHED(s);
        if (!wants(t2.x)) continue;
        const unsigned fv = ~(t2.y >= 0 ? cl.lfirst[t2.y] : NODEFIRST(t2.x));
        const int dense = dense_of(fv);
        if (in_lds) atomicOr((unsigned *)&bm_l[dense >> 5], 1u << (dense & 31)); else atomicOr(&bm_g[dense >> 5], 1u << (dense & 31));
      }
      __syncthreads();
      int Wsh = 0;
      while ((NT << Wsh) < nwords) Wsh++;
      const int W = 1 << Wsh;
      int cnt = 0;
      for (int x = 0; x < W; x++) { const int w = tid * W + x; if (w < nwords) cnt += __popc(in_lds ? bm_l[w] : bm_g[w]); }
      const int ex = block_excl_scan<NT>(sh, cnt);
      tpre[tid] = (unsigned)ex;
      __syncthreads();
      return Wsh;
    };
    auto rank_of = [&](int nb, int Wsh, int dense) -> int {
      const bool in_lds = ((nb + 31) >> 5) <= xw.bm_words;
      const int w = dense >> 5, tw = w >> Wsh;
      int r = (int)tpre[tw];
      for (int x = tw << Wsh; x < w; x++) r += __popc(in_lds ? bm_l[x] : bm_g[x]);
      r += __popc((in_lds ? bm_l[w] : bm_g[w]) & ((1u << (dense & 31)) - 1u));
      return r;
    };
    {
      auto dense1 = [&](unsigned fv) -> int { return dbase[fv >> s1] + (int)(fv & submask); };
      const int Wsh = rank_setup(nbits, n1, dense1, [](int) { return true; });
      unsigned mymax1 = ord(JAMD_LOG_ZERO), mymin1 = 0xffffffffu;
      for (int s = tid; s < n1; s += NT) {
        const int2 t2 = TOUCHED(s);
        const int node = t2.x, slot = t2.y;
        unsigned long long key; unsigned fvis;
        if (slot >= 0) {
          key = cl.lkey[slot]; fvis = ~cl.lfirst[slot];
          cl.lkey[slot] = 0ull; cl.lnode[slot] = -1; cl.lfirst[slot] = 0u;
        } else {
          key = atomicExch(&NODEKEY(node), 0ull);
          fvis = ~atomicExch(&NODEFIRST(node), 0u);
        }
        const int id = rank_of(nbits, Wsh, dense1(fvis));
        const unsigned vis = ~(unsigned)key;
        const int j = (int)(vis >> s1);
        const Tok tk = sv.load(j);
        const int4 nr = lx.node_b(node);
        Tok nw;
        nw.node = node; nw.pad0 = nr.x; nw.pad1 = tk.pad1;                 // (the forward-DFA state is inherited, :2120)
        nw.last_tre = tk.last_tre; nw.last_cword = tk.last_cword; nw.last_wid = tk.last_wid;
        nw.last_lscore = tk.last_lscore;
        if (!dfa && node != tk.node && nr.y != 0)                       // beam_intra_word_core() :2069-2082
          nw.last_lscore = max_successor_prob(lx, tk.last_cword, nr.y, memo) * lmw + pen;
        nw.score = unord((unsigned)(key >> 32));
        CUR(id) = nw;
        const unsigned kb = (unsigned)(key >> 32);
        CURKEY(id) = kb;
        mymax1 = max(mymax1, kb); mymin1 = min(mymin1, kb);
        { const int ti = TGT(node); if (ti >= 0) NODETOK(ti) = (unsigned)id + 1u; }
        ARR(id) = id;
      }
      atomicMax(&sh.maxbits, mymax1); atomicMin(&sh.minbits, mymin1);      // (the mid-frame sort's bins span them)
      __syncthreads();
    }
    if (TIMED && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[0] += n_ - tc; tc = n_; }
    // ---- M: the beam over the new tokens (:2774), the WHOLE array: step D starts from it.  exact_prune<FULL>: the sorted
    //         list + sweep replay give the extracted part, the sift replay below the extracted region the residual heap
    //         (beam_sweep.h; either direction); frames that machinery cannot hold run the extraction loop itself.
    int r_lo = 0, r_hi = n1 - 1;                             // tindex[r_lo..r_hi]: what the second half visits
    if (n1 > wk.beam) {
      const int k = wk.beam;
      exact_prune<WIDE, NT, true>(sh, &CURKEY(0), n1, k, Hlds, xw.heap_cap, Hglob, pm, welist, xw.prune_mode, Gcol, nullptr, &ARR(0));
      if (k < n1 - k) { r_lo = n1 - k; r_hi = n1 - 1; } else { r_lo = 0; r_hi = k - 1; }
      __syncthreads();
      clear_cells();                                         // the pruning step lay over the cell area
      __syncthreads();
    }
    if (tid == 0) { sh.maxbits = ord(JAMD_LOG_ZERO); sh.minbits = 0xffffffffu; }      // (step O collects them again)
    if (TIMED && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[3] += n_ - tc; tc = n_; }
    // ---- B0: word ends among tindex[r_lo..r_hi], in that order: save_trellis() :2209-2247
    const int atom_base = uni(sh.n_atom);
    int n_we = 0;
    for (int p0 = r_lo; p0 <= r_hi; p0 += NT) {
      const int p = p0 + tid;
      int id = -1, isend = 0;
      Tok tk;
      if (p <= r_hi) {
        id = ARR(p);
        tk = CUR(id);
        isend = (tk.pad0 >= 0 && !(tk.score < thr)) ? 1 : 0;
      }
      const int ex = block_excl_scan<NT>(sh, isend);
      if (isend) {
        const int w = n_we + ex;
        welist[w] = id;
        const int ai = atom_base + w;
        if (ai < wk.atom_cap) {
          jamd_trellis_atom a;
          a.wid = tk.pad0; a.last_tre = tk.last_tre; a.backscore = tk.score; a.lscore = tk.last_lscore;
          a.begintime = (short)((tk.last_tre < 0 ? -1 : ATOM(tk.last_tre).endtime) + 1);
          a.endtime = (short)(t - 1);
          ATOM(ai) = a;
        }
        if (!dfa && tk.pad0 != lx.tail_silwid && tk.score > JAMD_LOG_ZERO)       // wordend_best (:2307: no exit transition in multipath)
          atomicMax(&sh.we_best, ((unsigned long long)ordz(tk.score) << 32) | (unsigned)(~(unsigned)w));
      }
      n_we += sh.scan_total;
      __syncthreads();
    }
    if (tid == 0) { sh.n_atom = atom_base + n_we; sh.n_new = 0; }
    __syncthreads();
    if (sh.n_atom > wk.atom_cap) {
      if (tid == 0) res->status = JAMD_PASS1_OVERFLOW;
      stopped = true;
      __syncthreads();
      break;
    }
    if (final_call) break;                                   // :2803: the last call saves the word ends and stops
    unsigned long long tc4 = tc;
    (void)tc4;
#define MPTICK(i) do { if (TIMED && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[i] += n_ - tc4; tc4 = n_; } } while (0)
    MPTICK(4);

    // ---- B': cross-word transitions (:2779-2825), the roots expanded along their own arcs (:2467-2510).  The arcs come
    //         from lists flattened when the lexicon was created (jamd_lexicon::o_mp_*: one int4 {target, transition,
    //         root number * XW + transition number, root number | fscore bits} per transition a root really has, in
    //         visiting order): one load per candidate instead of the root's record, its node record and its arc.
    if (!wordmode && n_we > 0) {
      if (dfa) {
        const int nst = xw.n_mp_start, total = n_we * nst;
        for (int x = tid; x < total; x += NT) {
          const int w = x / nst, e = x - w * nst;
          const int4 ent = lx.at<int4>(xw.o_mp_start, e);
          const Tok tk = CUR(welist[w]);
          const int sword = tk.pad0;
          if (!lx.cat_pair(lx.wton(sword) * lx.ncat + lx.root_cat(ent.w))) continue;
          if (lx.nfwd && fwd_next(lx, tk.pad1, lx.root_cat(ent.w)) < 0) continue;      // forward DFA: no arc for this category (:2412-2422)
          const int last_word = lx.is_transparent(sword) ? tk.last_cword : sword;
          float tmpsum = tk.score;
          float ng = lx.penalty1;
          ng += (last_word >= 0) ? lx.cprob(last_word) : 0.0f;
          tmpsum += ng;
          xpush(sh, cl, ent.x, tmpsum + __int_as_float(ent.y), ((unsigned)w << s1) | (unsigned)ent.z);
        }
      } else {
        // beam_inter_word() :2296-2440: a word end is reduced to (score, LM context, rank) once; every (isolated root,
        // transition of the root) takes the best candidate and the earliest visit over the word ends
        const int niso = lx.isolatenum;
        lds_v4 *werec = (lds_v4 *)tpre;
        constexpr int kWeChunk = NT / 4;
        const int nexp = xw.n_mp_iso;
        for (int w0 = 0; w0 < n_we; w0 += kWeChunk) {
          const int nrec = min(kWeChunk, n_we - w0);
          if (tid < nrec) {
            const Tok tk = CUR(welist[w0 + tid]);
            const int sword = tk.pad0;
            const bool tr = lx.is_transparent(sword) != 0;
            const int last_word = tr ? tk.last_cword : sword;
            const bool trans2 = tr && tk.last_cword >= 0 && lx.is_transparent(tk.last_cword);
            u32x4 rec;
            rec.x = __float_as_uint(tk.score); rec.y = (unsigned)(last_word < 0 ? -1 : lx.wton(last_word));
            rec.z = (unsigned)(w0 + tid) | (trans2 ? 0x80000000u : 0u);
            rec.w = (sword == lx.tail_silwid) ? 1u : 0u;                           // the sentence-final silence is followed by nothing
            werec[tid] = rec;
          }
          __syncthreads();
          int parts = NT / (nexp > 0 ? nexp : 1);                // the word ends of the chunk are split over the idle threads
          if (parts < 1) parts = 1;
          if (parts > nrec) parts = nrec;
          const int total = nexp * parts;
          for (int y = tid; y < total; y += NT) {
            const int part = y / nexp, e = y - part * nexp;
            const int4 ent = lx.at<int4>(xw.o_mp_iso, e);
            const int i = ent.w;
            const float trn = __int_as_float(ent.y);
            const int wn = lx.iwtab ? 0 : lx.iso_root(i).y;
            unsigned long long best = 0ull; unsigned nfirst = 0u;
            // four word ends at a time: their table reads go out together (a maximum and a minimum do not care about the order)
            constexpr int G = 4;
            for (int w0g = part; w0g < nrec; w0g += G * parts) {
              u32x4 rec[G]; float p[G]; bool live[G];
#pragma unroll
              for (int g = 0; g < G; g++) {
                const int wv = w0g + g * parts;
                live[g] = wv < nrec;
                rec[g] = werec[live[g] ? wv : part];
                live[g] = live[g] && rec[g].w == 0u;
                const int ctx = (int)rec[g].y;
                p[g] = (!live[g] || ctx < 0) ? 0.0f
                       : lx.iwtab ? lx.iwtab[(size_t)ctx * niso + i]
                       : bigram_prob(lx, ctx, lx.wton(wn)) + lx.cprob(wn);
              }
#pragma unroll
              for (int g = 0; g < G; g++) {
                if (!live[g]) continue;
                float tmpsum = __uint_as_float(rec[g].x);
                const float ng = p[g] * lmw + pen;
                tmpsum += ng;
                if (rec[g].z & 0x80000000u) tmpsum += lx.lm_penalty_trans;
                const float cand = tmpsum + trn;
                if (cand <= JAMD_LOG_ZERO) continue;
                const unsigned nv = ~(((rec[g].z & 0x7fffffffu) << s1) | (unsigned)ent.z);
                const unsigned long long key = ((unsigned long long)ordz(cand) << 32) | nv;
                if (key > best) best = key;
                if (nv > nfirst) nfirst = nv;
              }
            }
            if (best != 0ull) xpush_key(sh, cl, ent.x, best, nfirst);
          }
          __syncthreads();
        }
        MPTICK(5);
        if (sh.we_best != 0ull) {                       // beam_inter_word_factoring() :2549-2637
          const unsigned long long kb = sh.we_best;
          const float best_score = unord((unsigned)(kb >> 32));
          const Tok tk = CUR(welist[(int)(~(unsigned)kb)]);
          const int sword = tk.pad0;
          const bool trans2 = lx.is_transparent(sword) && tk.last_cword >= 0 && lx.is_transparent(tk.last_cword);
          const int nsh = xw.n_mp_shared;
          for (int e = tid; e < nsh; e += NT) {
            const int4 ent = lx.at<int4>(xw.o_mp_shared, e);
            const float ng = __int_as_float(ent.w) * lmw + pen;
            float tmpsum = best_score;
            tmpsum += ng;
            if (trans2) tmpsum += lx.lm_penalty_trans;
            if (tmpsum < thr) continue;                                           // :2580
            xpush(sh, cl, ent.x, tmpsum + __int_as_float(ent.y), ((unsigned)n_we << s1) | (unsigned)ent.z);
          }
        }
      }
    }
    __syncthreads();
    MPTICK(6);
    // ---- C2: the candidates of the second half: a node that holds a token of the first half is improved in place
    //          (propagate_token() :1945: strictly better only), the others are appended in creation order
    const int n2 = uni(sh.n_new);
    int n_tot = n1;
    if (n2 > 0) {
      if (n1 + n2 > wk.tok_cap) {
        if (tid == 0) res->status = JAMD_PASS1_OVERFLOW;
        stopped = true;
        __syncthreads();
        break;
      }
      const int nbits2 = (n_we + 1) * slots2 + (dfa ? 0 : lx.nshared * XW);
      auto dense2 = [&](unsigned fv) -> int { return (int)(fv >> s1) * slots2 + (int)(fv & submask); };
      const int Wsh = rank_setup(nbits2, n2, dense2, [&](int node) { return NODETOK(TGT(node)) == 0u; });
      if (tid == 0) sh.n_arc = 0;
      __syncthreads();
      for (int s = tid; s < n2; s += NT) {
        const int2 t2 = TOUCHED(s);
        const int node = t2.x, slot = t2.y;
        unsigned long long key; unsigned fvis;
        if (slot >= 0) {
          key = cl.lkey[slot]; fvis = ~cl.lfirst[slot];
          cl.lkey[slot] = 0ull; cl.lnode[slot] = -1; cl.lfirst[slot] = 0u;
        } else {
          key = atomicExch(&NODEKEY(node), 0ull);
          fvis = ~atomicExch(&NODEFIRST(node), 0u);
        }
        const int ti = TGT(node);                          // (>= 0: the candidates of the second half come from the roots' lists)
        const unsigned have = NODETOK(ti);
        const float cand = unord((unsigned)(key >> 32));
        int id;
        if (have != 0u) {
          id = (int)have - 1;
          if (!(CUR(id).score < cand)) continue;
        } else {
          id = n1 + rank_of(nbits2, Wsh, dense2(fvis));
          atomicAdd(&sh.n_arc, 1);
        }
        const unsigned vis = ~(unsigned)key;
        const int w = (int)(vis >> s1), sub = (int)(vis & submask), ri = sub / XW;
        const bool fact = w == n_we;
        const Tok src = CUR(welist[fact ? (int)(~(unsigned)sh.we_best) : w]);
        const int sword = src.pad0;
        const int last_word = lx.is_transparent(sword) ? src.last_cword : sword;
        float ls;
        if (dfa) {                                           // beam_inter_word() :2452-2461
          ls = lx.penalty1;
          ls += (last_word >= 0) ? lx.cprob(last_word) : 0.0f;
        } else if (!fact) {                                  // :2430-2438
          float p = 0.0f;
          if (last_word >= 0) {
            if (lx.iwtab) p = lx.iwtab[(size_t)lx.wton(last_word) * lx.isolatenum + ri];
            else { const int wn = lx.iso_root(ri).y; p = bigram_prob(lx, lx.wton(last_word), lx.wton(wn)) + lx.cprob(wn); }
          }
          ls = p * lmw + pen;
        } else {                                             // beam_inter_word_factoring() :2572-2573
          ls = lx.shared_root(ri).y * lmw + pen;
        }
        Tok nw;
        nw.node = node; nw.pad0 = lx.node_b(node).x;
        nw.pad1 = (dfa && lx.nfwd) ? fwd_next(lx, src.pad1, lx.root_cat(lx.startnum - 1 - ri)) : 0;   // the arc step B' found (:2415-2420; roots are numbered from startnum-1 down)
        nw.last_tre = atom_base + (fact ? (int)(~(unsigned)sh.we_best) : w); nw.last_cword = last_word; nw.last_wid = sword;
        nw.last_lscore = ls; nw.score = cand;
        CUR(id) = nw;
        if (have == 0u) { NODETOK(ti) = (unsigned)id + 1u; ARR(id) = id; }
      }
      __syncthreads();
      n_tot = n1 + uni(sh.n_arc);
    }
    if (n_tot > max_tokens) max_tokens = n_tot;
    if (pm.pstat && tid == 0 && n1 > wk.beam) pm.pstat[12] += 1;             // frames whose new tokens exceeded the beam
    if (pm.pstat && tid == 0) { pm.pstat[8] += n_tot; pm.pstat[9] += n_surv; pm.pstat[10] += n_we; pm.pstat[11] += 1; }   // work counters (jamd_beam_prune_stats())
    MPTICK(7);
#undef MPTICK
    if (TIMED && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[1] += n_ - tc; tc = n_; }
    // ---- O: output probabilities on emitting nodes (:2930-2943); nodetok[] is emptied on the way
    {
      const XRowRef row{scores + (size_t)(t_begin + t - base) * S, rowc, wk.row_cache != 0};
      unsigned mymax = ord(JAMD_LOG_ZERO), mymin = 0xffffffffu, myemax = ord(JAMD_LOG_ZERO);
      for (int i = tid; i < n_tot; i += NT) {
        const Tok tk = CUR(i);
        { const int ti = TGT(tk.node); if (ti >= 0) NODETOK(ti) = 0u; }
        const int4 nr = lx.node_b(tk.node);
        float sc = tk.score;
        if (nr.w != JAMD_AS_NONE) {
          const int ent = outprob_entry(lx, nr.w, nr.z, tk.last_wid);
          sc += ent >= 0 ? row[ent]
                         : cd_reduce(row, lx.set_states_ptr(), lx.set_off(~ent), lx.set_off(~ent + 1), lx.cdset_method, lx.cdmax_num);
          CUR(i).score = sc;
          myemax = max(myemax, ordz(sc));
        }
        const unsigned b = ordz(sc);
        CURKEY(i) = b;
        mymax = max(mymax, b); mymin = min(mymin, b);
      }
      atomicMax(&sh.maxbits, mymax); atomicMin(&sh.minbits, mymin); atomicMax(&sh.emaxbits, myemax);
    }
    __syncthreads();
    row_request(t + 1);
    if (TIMED && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[2] += n_ - tc; tc = n_; }
    {
      const float mx = unord(sh.emaxbits);
      thr = (wk.width >= 0.0f) ? (mx - wk.width) : JAMD_LOG_ZERO;
    }
    if (n_tot == 0) {
      if (tid == 0) { res->status = JAMD_PASS1_DIED; res->died_at = t; }
      stopped = true;
      __syncthreads();
      break;
    }
    // ---- D: the frame's final cut over tindex[] as step M left it plus the appended tokens
    for (int p = tid; p < n_tot; p += NT) KEY2(p) = CURKEY(ARR(p));
    __syncthreads();
    const int n_keep = exact_prune<WIDE, NT>(sh, &KEY2(0), n_tot, wk.beam, Hlds, xw.heap_cap, Hglob, pm, welist, xw.prune_mode, Gcol, nullptr);
    for (int j = tid; j < n_keep; j += NT) sv.store(j, CUR(ARR(welist[j])));
    if (tid == 0) sh.n_surv = n_keep;
    clear_cells();
    __syncthreads();
    if (TIMED && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[3] += n_ - tc; tc = n_; }
  }
  __syncthreads();

  if (smode == 1) {
    if constexpr (!WIDE) {
      if (!stopped) {
        u32x4 *dst = (u32x4 *)(ub + wk.o_sv);
        for (int i = tid; i < wk.sv_bytes / 16; i += NT) dst[i] = sv.p[i];
      }
    }
    if (tid == 0) {
      ss->started = 1; ss->active = stopped ? 0 : 1; ss->frames_done = T; ss->n_surv = sh.n_surv; ss->thr = thr;
      ss->n_atom = sh.n_atom; ss->ties = 0; ss->ties_we = 0; ss->ties_cut = 0;
      ss->max_tokens = max_tokens;
      res->natom = min(sh.n_atom, wk.atom_cap); res->frames = T; res->max_tokens = max_tokens;
      res->ties = 0;
      if (TIMED) for (int i = 0; i < 8; i++) res->phase_us[i] += (int)(ph[i] / 100ull);
      if (pstat_glob) for (int i = 0; i < 16; i++) pstat_glob[i] += sh.pst[i];
    }
    return;
  }
  if (ss && tid == 0) { ss->active = 0; ss->started = 1; ss->frames_done = T; }

  // ---- find_1pass_result() :399-455 + trace_backptr() :294-340
  const int natom = min(sh.n_atom, wk.atom_cap);
  if (tid == 0) sh.best_atom = -1;
  __syncthreads();
  if (res->status == JAMD_PASS1_OK && dfa) {
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
    for (int i = tid; i < natom; i += NT)
      if (kb != 0ull && ATOM(i).endtime == lt && (unsigned)ATOM(i).wid == 0xffffffffu - (unsigned)kb &&
          ord(ATOM(i).backscore) == (unsigned)(kb >> 32)) sh.best_atom = i;
  } else if (res->status == JAMD_PASS1_OK) {
    int bt = -1;
    for (int i = tid; i < natom; i += NT)
      if (ATOM(i).wid == lx.tail_silwid && ATOM(i).backscore > JAMD_LOG_ZERO && ATOM(i).endtime > bt) bt = ATOM(i).endtime;
    if (tid == 0) sh.n_arc = -1;
    __syncthreads();
    if (bt >= 0) atomicMax(&sh.n_arc, bt);
    __syncthreads();
    bt = sh.n_arc;
    for (int i = tid; i < natom; i += NT)
      if (bt >= 0 && ATOM(i).wid == lx.tail_silwid && ATOM(i).backscore > JAMD_LOG_ZERO && ATOM(i).endtime == bt) sh.best_atom = i;
  }
  __syncthreads();
  if (tid == 0) {
    res->natom = natom; res->ties = 0; res->max_tokens = max_tokens;
    res->ties_node = 0; res->ties_wordend = 0; res->ties_cut = 0;
    if (pstat_glob) for (int i = 0; i < 16; i++) pstat_glob[i] += sh.pst[i];
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
#undef NODETOK
#undef TGT
#undef ARR
#undef KEY2
}
