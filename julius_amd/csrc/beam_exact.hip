// beam_exact.hip -- first-pass token passing with the REFERENCE'S TIE SEMANTICS, frame-parallel (K6x).
//
// beam.hip's beam_pass1_kernel resolves exact score ties canonically (larger source id, smaller node on
// the rank cut); the reference resolves them by its visiting order:
//   * propagate_token() (libjulius/src/beam.c:1945-1980) replaces a token only on a STRICTLY better score,
//     so among equal candidates the one visited first wins;
//   * beam_inter_word() keeps the first of equally good word ends as wordend_best (:2308);
//   * the visiting order of a frame is tindex[n_start..n_end] as sort_token_no_order() (:1492) left it:
//     creation order when nothing is pruned, else the output of a partial heap sort
//     (sort_token_upward / _downward, :1342-1480) over the tokens in creation order;
//   * creation order is the order of first visits (create_token(), :1148).
// This kernel reproduces all of that with the whole workgroup:
//   1. every candidate carries its visiting index vis = (position j of the source in the visiting order,
//      transition number within the source: self, next, extra arcs in wchmm order, then the roots from
//      startnum-1 down to 0; the factoring pass of beam_inter_word_factoring() counts as source n_surv).
//      The Viterbi cell is atomicMax(score bits || ~vis): best score, earliest visit -- first-writer-wins.
//      A second atomicMax(~vis) per cell keeps the node's FIRST visit.
//   2. creation order = rank of the first visit: one bit per visiting index in a bitmap, prefix popcount.
//   3. rank pruning = the reference's heap, exactly:
//        - heapify runs level-parallel (the sift-downs of one tree level touch disjoint subtrees and the
//          reference runs the levels bottom-up, so the result is the sequential one), with the levels
//          overlapped inside a wave (heapify_overlapped());
//        - the extraction loop of sort_token_upward() is replaced by its closed form.  While the element
//          taken from the tail is smaller than every element still to be extracted, an extraction is a
//          hole running down the path of larger children (left on ties): the heap is a tree of stable
//          merges, and the extraction order is (score descending, PRE-ORDER index of the heap position
//          ascending).  The exceptions ("events": the tail element is itself among the top k) re-insert
//          that element at the end of the current max path; they are rare (a few per frame), found and
//          replayed one by one by a single wave with range queries over the top-k list (sorted by counting
//          over score bins), and only up to the last turn that can still change the order.  The
//          equivalence was fuzzed against the sequential code (tests/test_prune_order.py does it on the
//          device; HISTORY.md part II section 3 "K6x" has the argument).
//        - sort_token_downward() (beam < tokens <= 2 beam) and oversize frames run the sequential
//          extraction on one lane, in LDS.
// Everything else -- LM factoring, outprob_style(), trellis atoms, score pruning -- is the arithmetic of
// beam_pass1_kernel.  The word trellis equals the reference's bit for bit, ties included
// (tests/test_beam_gpu.py::test_exact_*).  N-gram, grammar and word-list lexicons; non-multipath.
#include <type_traits>
#include "beam_common.h"
#include "beam_exact.h"

namespace {
using namespace jamdb;

#ifndef JAMD_XBEAM_CB
#define JAMD_XBEAM_CB 4                 // tokens per thread carried together through the finalize step
#endif
constexpr int kMaxL = 20;                // heap positions < 2^21
// The instrumented instantiation (JAMD_BEAM_TIMING=1) reports the four steps of a frame and the four parts of the
// pruning step in phase_us[0..7].  Finer probes exist only in development builds (-DJAMD_DEV, tools/build_variant.sh,
// tools/exact_probe.sh): JAMD_XBEAM_PROBE = 1 / 2 / 3 / 5 puts the sub-step clocks of steps 0-B / step C / the event
// replay / heap fill + heapify into phase_us[4..7] instead; 4 reports the shader clock (MHz) in phase_us[7].
#if !defined(JAMD_DEV) || !defined(JAMD_XBEAM_PROBE)
#undef JAMD_XBEAM_PROBE
#define JAMD_XBEAM_PROBE 0
#endif

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct XRowRef {                 // this frame's score row: its LDS copy or the row in global memory
  const float *g; const lds_f32 *l; bool lds;
  __device__ __forceinline__ float operator[](int i) const { return lds ? l[i] : g[i]; }
};

struct XShared {
  unsigned long long we_best;            // (ord(score + wordend_a), ~j): best word end, earliest visit
  int n_new, n_we, n_arc, n_atom, n_surv, best_atom, nB, i_last;
  unsigned maxbits, minbits;
  unsigned sel_digit, sel_need, sel_count;
  unsigned wsum[NT / 64], wsum2[NT / 64];
  int scan_total, scan_total2;
  int sw_nev, sw_limit, sw_fail, sw_changed, sw_ncl;     // the sweep replay (beam_sweep.h)
  int sw_ticks, sw_nev_out, sw_prof[8];
  int pst[16];                                   // this launch's share of jamd_beam_prune_stats(): kept here, added to the slice once at the end                      // its duration (100 MHz ticks), events held at the end
  int sw_info;                                   // last pruning step: rounds of the sweep replay, -1 = it gave up, 0 = not used
  int df_prof[4];                                // down_finish(): load, dependencies, sifts, output (100 MHz ticks; development)
  unsigned emaxbits;                             // multipath frame: best score among the tokens on emitting nodes (the score-pruning envelope)
  unsigned long long ph[8];                      // phase clocks of the instrumented instantiation (JAMD_BEAM_TIMING=1)
};

struct XCells {
  unsigned char *ub; unsigned o_nodekey, o_nodefirst, o_touched;
  lds_u64 *lkey; lds_i32 *lnode; lds_u32 *lfirst;
  int nslot;
};
#ifndef JAMD_XPROBES
#define JAMD_XPROBES 24
#endif
constexpr int kXProbes = JAMD_XPROBES;
// The probe loop of a cell insert: fully unrolled into 24 nested conditionals (0: the compiler's choice and the default) or
// kept as ONE loop (1).  Unrolled, every level saves an execution mask and a condition mask, eight call sites deep -- 370
// of the kernel's remaining scalar spill slots and a sixth of its code -- but those levels only RUN for the rare lane
// that probes that far, while the rolled loop pays its mask bookkeeping on every insert: measured on one box, round 5
// (profiles/r05b_ab_register_diet.txt), rolled is 2.5 - 3 % slower on every configuration (C3 512 utterances 208.0 vs
// 203.0 ms, C4 946.6 vs 930.7 ms), as its round-3 predecessor was.  Static spill counts are not run time.
#ifndef JAMD_XPROBE_ROLLED
#define JAMD_XPROBE_ROLLED 0
#endif

// The thread index as a value the optimiser cannot carry from one frame to the next.  Everything derived from it (lane
// and wave numbers, per-thread addresses into a dozen arrays) is loop-invariant over the frame loop; hoisted, those
// values cost more registers than the kernel has and come back from scratch memory in the middle of serial sections.
// Recomputing them where they are used is a few VALU instructions.
__device__ __forceinline__ int tid_now() {
  int t = (int)threadIdx.x;
  asm volatile("" : "+v"(t));
  __builtin_assume(t >= 0 && t < 1024);
  return t;
}

// block-wide exclusive scan of one int per thread (two barriers); total in sh.scan_total
template <int NT>
__device__ __forceinline__ int block_excl_scan(XShared &sh, int v) {
  const int tx = tid_now(), lane = tx & 63, wv = tx >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) sh.wsum[wv] = (unsigned)incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wv; w++) base += (int)sh.wsum[w];
  if (tx == NT - 1) sh.scan_total = base + incl;
  __syncthreads();
  return base + incl - v;
}

// the same for two ints per thread (totals in sh.scan_total / sh.scan_total2)
template <int NT>
__device__ __forceinline__ void block_excl_scan2(XShared &sh, int a, int b, int &ea, int &eb) {
  const int tx = tid_now(), lane = tx & 63, wv = tx >> 6;
  int ia = a, ib = b;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int oa = __shfl_up(ia, off, 64), ob = __shfl_up(ib, off, 64);
    if (lane >= off) { ia += oa; ib += ob; }
  }
  if (lane == 63) { sh.wsum[wv] = (unsigned)ia; sh.wsum2[wv] = (unsigned)ib; }
  __syncthreads();
  int ba = 0, bb = 0;
  for (int w = 0; w < wv; w++) { ba += (int)sh.wsum[w]; bb += (int)sh.wsum2[w]; }
  if (tx == NT - 1) { sh.scan_total = ba + ia; sh.scan_total2 = bb + ib; }
  __syncthreads();
  ea = ba + ia - a; eb = bb + ib - b;
}

// A value every lane of the wave holds alike, moved to a scalar register: the compiler cannot know that a value read
// from LDS (or passed to a function that is not inlined) is uniform, and would run the loops it controls under
// execution masks.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
template <typename T>
__device__ __forceinline__ T JAMD_LDS *uni(T JAMD_LDS *p) {
  return (T JAMD_LDS *)(unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)p);
}

// the LDS operations of one wave execute in order: this only keeps the compiler from moving them across
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned ordz(float f) { return ord(f + 0.0f); }   // -0.0 and +0.0 compare equal as floats

// Candidates for one node: key = the best of them (score bits || ~visiting index), nfirst = ~(their earliest
// visiting index).  propagate_token() :1945 with the visiting index as the tie breaker.
__device__ __forceinline__ void xpush_key(XShared &sh, const XCells &cl, int node, unsigned long long key, unsigned nfirst) {
  bool first = false;
  int slot = -1;
  if (cl.nslot > 0) {
    unsigned h = __umulhi((unsigned)node * 2654435761u, (unsigned)cl.nslot);
#if JAMD_XPROBE_ROLLED
#pragma nounroll
#endif
    for (int pr = 0; pr < kXProbes; pr++) {
      const int o = atomicCAS((int *)&cl.lnode[h], -1, node);
      if (o == -1 || o == node) { slot = (int)h; first = (o == -1); break; }
      h = (h + 1 == (unsigned)cl.nslot) ? 0u : h + 1;
    }
  }
  if (slot >= 0) {
    atomicMax((unsigned long long *)&cl.lkey[slot], key);
    atomicMax((unsigned *)&cl.lfirst[slot], nfirst);
  } else {
    const unsigned long long old =
        atomicMax(reinterpret_cast<unsigned long long *>(cl.ub + (unsigned)(cl.o_nodekey + 8u * (unsigned)node)), key);
    atomicMax(reinterpret_cast<unsigned *>(cl.ub + (unsigned)(cl.o_nodefirst + 4u * (unsigned)node)), nfirst);
    first = (old == 0ull);
  }
  const int s = wave_alloc(&sh.n_new, first);
  if (first) *reinterpret_cast<int2 *>(cl.ub + (unsigned)(cl.o_touched + 8u * (unsigned)s)) = make_int2(node, slot);
}
// one candidate
__device__ __forceinline__ void xpush(XShared &sh, const XCells &cl, int node, float score, unsigned vis) {
  if (score <= JAMD_LOG_ZERO) return;
  xpush_key(sh, cl, node, ((unsigned long long)ordz(score) << 32) | (unsigned)(~vis), ~vis);
}

// ---- rank pruning with the reference's heap ----------------------------------------------------------
// pre-order key of heap position p (1-based): bit string of p below its leading one, left aligned, then
// the depth -- an ancestor sorts before its descendants, a left subtree before the right one
__device__ __forceinline__ unsigned prekey(unsigned p) {
  const int L = 31 - __clz((int)p);
  return (((p - (1u << L)) << (kMaxL - L)) << 5) | (unsigned)L;
}
__device__ __forceinline__ unsigned prekey_pos(unsigned key) {
  const int L = (int)(key & 31u);
  return (1u << L) + ((key >> 5) >> (kMaxL - L));
}
// is heap position p inside the subtree of position c?  (p == 0: nowhere)
__device__ __forceinline__ bool insub(unsigned p, unsigned c) {
  if (p < c) return false;
  const int d = __clz((int)c) - __clz((int)p);
  return (p >> d) == c;
}

struct PruneMem {                // LDS regions of the pruning step (they overlay the empty Viterbi cells)
  lds_u64 *compR, *compT;        // [b_cap] each: the sorted top list (score bits << 32 | ~prekey at collection time), and scratch for sorting it
  lds_u32 *vposR;                // [b_cap] current virtual heap position per rank
  lds_u32 *idR;                  // [b_cap] token id per rank
  lds_u32 *idT;                  // [b_cap] wide layout: token ids beside compT while the list is being sorted
  lds_u32 *hist;                 // [2048]
  lds_u32 *tailmask;             // [(beam + 31) / 32 + 1]
  lds_i32 *cand;                 // [kMaxCand] tail candidates in the order of their turns: extraction index i
  lds_i32 *occ;                  // [kMaxCand] rank of the element at the candidate's tail position, -1 = none
  lds_i32 *need;                 // [kMaxCand + 4] 1 = (re)scan wanted; [kMaxCand..] = ncand, cursor, finished
  lds_i32 *takers;               // [kMaxCand + 1][kTakers + 1] chain occupants per candidate (+ their count); last row: serial form
  lds_i32 *ordv;                 // [kMaxCand] candidate slots in the order of their turns
  int b_cap;
  unsigned char JAMD_LDS *sw_region;   // the sweep replay (beam_sweep.h): all of the pruning step's overlay, laid out afresh
  int sw_bytes;
  unsigned char *sw_glob;        // its global scratch (sweep_global_bytes()), nullptr = no sweep
  int *pstat;                    // [16] how the pruning steps of this utterance were resolved (XShared::pst), or nullptr
};

template <bool UP, typename HP>
__device__ __forceinline__ void heap_sift(HP H, int n, int parent, unsigned long long s) {
  const unsigned sv = (unsigned)(s >> 32);
  int child;
  while ((child = parent * 2) <= n) {
    unsigned long long c = H[child];
    if (child < n) {
      const unsigned long long c2 = H[child + 1];
      const unsigned a = (unsigned)(c >> 32), b = (unsigned)(c2 >> 32);
      if (UP ? (a < b) : (a > b)) { child++; c = c2; }
    }
    const unsigned cv = (unsigned)(c >> 32);
    if (UP ? (sv >= cv) : (sv <= cv)) break;
    H[parent] = c;
    parent = child;
  }
  H[parent] = s;
}

// first loop of sort_token_upward/_downward (:1354-1367): level-parallel
template <bool UP, int NT, typename HP>
__device__ __forceinline__ void heapify_levels(HP H, int n) {
  const int top = n / 2;
  if (top >= 1) {
    for (int L = 31 - __clz(top); L >= 0; L--) {
      const int lo = 1 << L, hi = min((2 << L) - 1, top);
      for (int root = lo + tid_now(); root <= hi; root += NT) heap_sift<UP>(H, n, root, H[root]);
      __syncthreads();
    }
  }
}

// The same loop for a heap in LDS, with the levels overlapped.  A sift-down that starts at depth L is at depth L + t
// after t steps, where it reads the two children below and writes its own level; the sift that started one level
// further down wrote that children's level one step earlier IF it started one step earlier.  So the sifts of
// different levels can run together, one step apart (deepest level first), as long as a step's reads come before
// its writes and see the writes of the step before -- which is how the lanes of ONE wave execute.  Each wave takes
// four of the 64 subtrees rooted at depth 6 (several sifts per lane, their LDS reads in flight together); after one
// workgroup barrier wave 0 finishes depths 5..0 the same way.  A heap of 3 000 entries takes 10 + 17 dependent steps
// and one barrier instead of 66 steps and 11 barriers; the result is the sequential one (every sift reads
// exactly the values it would read in the reference's order).
struct SiftSlot { int parent, t0; unsigned long long s; bool live; };

template <bool UP, int R>
__device__ __forceinline__ void sift_overlapped(lds_u64 *H, int n, SiftSlot (&sl)[R], int gsteps) {
  for (int g = 0; g < gsteps; g++) {
    u32x4 ch[R]; bool run[R], leaf[R];
#pragma unroll
    for (int r = 0; r < R; r++) {                              // reads of this step
      run[r] = sl[r].live && g >= sl[r].t0;
      leaf[r] = 2 * sl[r].parent > n;
      ch[r] = u32x4{0u, 0u, 0u, 0u};
      if (run[r] && !leaf[r]) ch[r] = *(const lds_v4 *)&H[2 * sl[r].parent];
    }
    bool any = false;
#pragma unroll
    for (int r = 0; r < R; r++) {                              // decisions and writes
      if (!run[r]) { any |= sl[r].live; continue; }
      const int child = 2 * sl[r].parent;
      const unsigned a = ch[r].y, b = ch[r].w, sv = (unsigned)(sl[r].s >> 32);
      const bool right = child < n && (UP ? (a < b) : (a > b));
      const unsigned cv = right ? b : a;
      if (leaf[r] || (UP ? (sv >= cv) : (sv <= cv))) { H[sl[r].parent] = sl[r].s; sl[r].live = false; }
      else {
        H[sl[r].parent] = right ? (((unsigned long long)ch[r].w << 32) | ch[r].z) : (((unsigned long long)ch[r].y << 32) | ch[r].x);
        sl[r].parent = child + (right ? 1 : 0);
        any = true;
      }
    }
    wave_sync();
    if (!__any(any)) break;
  }
}

constexpr int kSplitLevel = 6;           // depths >= 6: 64 subtrees, 64 / (waves of the workgroup) per wave; depths < 6: wave 0
template <bool UP, int R, int NT>
__device__ __forceinline__ void heapify_subtrees(lds_u64 *H, int n, int Ltop, int Lmax) {
  constexpr int kSubPerWave = 64 / (NT / 64);
  const int tx = tid_now(), lane = tx & 63, wv = tx >> 6;
  const int D = Ltop - kSplitLevel + 1, top = n / 2;       // a subtree has 2^D - 1 roots: index 1 .. 2^D - 1 inside it
  SiftSlot sl[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int idx = r * 64 + lane;                          // R * 64 = kSubPerWave << D
    const int sub = idx >> D, within = idx & ((1 << D) - 1);
    const int d = within ? 31 - __clz(within) : 0;
    const int pos = (((1 << kSplitLevel) + kSubPerWave * wv + sub) << d) + (within - (1 << d));
    sl[r].live = within != 0 && sub < kSubPerWave && pos <= top;
    sl[r].parent = pos; sl[r].t0 = Ltop - (kSplitLevel + d);
    sl[r].s = sl[r].live ? H[pos] : 0ull;
  }
  sift_overlapped<UP, R>(H, n, sl, (Ltop - kSplitLevel) + (Lmax - kSplitLevel + 1));
}

// returns false when the heap is too deep for the register slots (the caller runs heapify_levels)
template <bool UP, int NT>
__device__ __forceinline__ bool heapify_overlapped(lds_u64 *H, int n) {
  const int top = n / 2;
  if (top < 1) return true;
  const int Ltop = 31 - __clz(top), Lmax = 31 - __clz(n);
  if (Ltop >= kSplitLevel) {
    const int D = Ltop - kSplitLevel + 1;
    constexpr int R5 = ((64 / (NT / 64)) << 5) / 64;      // register slots a lane needs at D = 5
    if (D <= 5) heapify_subtrees<UP, R5, NT>(H, n, Ltop, Lmax);
    else if (D == 6) heapify_subtrees<UP, 2 * R5, NT>(H, n, Ltop, Lmax);
    else if (D == 7 && R5 <= 2) heapify_subtrees<UP, (R5 <= 2 ? 4 * R5 : 1), NT>(H, n, Ltop, Lmax);
    else return false;
    __syncthreads();
  }
  const int tx = tid_now();
  if (tx < 64) {
    const int r = tx + 1, L = 31 - __clz(r);
    const int Lt = Ltop < kSplitLevel ? Ltop : kSplitLevel - 1;
    SiftSlot sl[1];
    sl[0].live = r <= top && L <= Lt;
    sl[0].parent = r; sl[0].t0 = Lt - L;
    sl[0].s = sl[0].live ? H[r] : 0ull;
    sift_overlapped<UP, 1>(H, n, sl, Lt + (Lmax + 1));
  }
  __syncthreads();
  return true;
}

// second loop (:1368-1383) on one lane
template <bool UP, typename HP>
__device__ __forceinline__ void heap_extract_serial(HP H, int n, int cnt) {
  int m = n;
  while (m > n - cnt) {
    const unsigned long long s = H[m];
    H[m] = H[1];
    m--;
    heap_sift<UP>(H, m, 1, s);
  }
}

// The same loop PIPELINED on one wave (heap in LDS).  Extraction e takes the tail element s = H[n - e + 1], puts the
// root there and lets s run down from the root; its step at depth d reads the two children at depth d + 1 and writes
// depth d.  Extraction e + 1 may therefore start two steps behind extraction e: every value it reads has received all
// earlier extractions' writes one step before at the latest (a step's reads come before its writes, and a step sees
// the writes of the step before -- the lanes of one wave).  One more dependency: the tail position n - e + 1 itself
// lies inside the heaps of the earlier extractions, which may still read it, move it up or end there; that is
// possible only for an extraction whose hole is an ancestor (or the position itself), and e waits until no such
// extraction is in flight.  Lane (e - 1) & 63 runs extraction e; about depth / 2 extractions are in flight, so the
// loop costs ~2 steps an extraction instead of one per level: 7x at 4000 extractions from 8000 tokens -- the form
// sort_token_downward() and the fall-backs of the closed-form extraction run in.  Reads and writes are the
// sequential loop's, so is the result.
template <bool UP>
__device__ __noinline__ void heap_extract_pipelined(lds_u64 *H, int n, int cnt) {
  // Branch-free body: idle lanes (p == 0) read H[0..1] and write H[0] (the unused slot in front of the heap), every lane
  // reads the starting extraction's tail element and the root (a broadcast), so no execution-mask juggling is left --
  // a single wave pays ~8 cycles per instruction, the count is what matters.
  H = uni(H); n = uni(n); cnt = uni(cnt);
  const int lane = threadIdx.x & 63;
  lds_u32 *H32 = (lds_u32 *)H;
  int p = 0, m = 0;                                              // p == 0: the lane is idle
  unsigned shi = 0u, slo = 0u;                                   // the element on its way down
  int e = 1;
  bool rest = false;                                             // a start in the previous step: this step starts nothing
  for (;;) {
    // may extraction e start in this step?
    bool st = false;
    const int q = n - e + 1;
    if (!rest && e <= cnt) st = __ballot(p != 0 && insub((unsigned)q, (unsigned)p)) == 0ull;
    rest = st;
    const bool mine = st && lane == ((e - 1) & 63);
    if (mine) { p = 1; m = n - e; }
    const int qe = st ? q : 0;
    // reads
    const unsigned long long tail = H[qe], root = H[1];
    const u32x4 ch = *(const lds_v4 *)&H[2 * p];
    wave_sync();
    // decisions and writes
    if (mine) { shi = (unsigned)(tail >> 32); slo = (unsigned)tail; }
    H[qe] = root;                                                // (no start: slot 0)
    const int child = 2 * p;
    const bool inner = p != 0 && child <= m;
    const bool right = inner && child < m && (UP ? (ch.y < ch.w) : (ch.y > ch.w));
    const unsigned cv = right ? ch.w : ch.y, cl = right ? ch.z : ch.x;
    const bool stop = !inner || (UP ? (shi >= cv) : (shi <= cv));
    H32[2 * p] = stop ? slo : cl;
    H32[2 * p + 1] = stop ? shi : cv;
    p = stop ? 0 : child + (right ? 1 : 0);
    wave_sync();
    e += st ? 1 : 0;
    if (e > cnt && __ballot(p != 0) == 0ull) break;
  }
}

// k-th largest of the score bits in H[1..n] (radix select, 11 bits a pass over the bits in which the
// frame's max and min differ).  Returns the value; all threads.
template <int NT, typename HP>
__device__ __forceinline__ unsigned kth_largest(XShared &sh, HP H, int n, int k, lds_u32 *hist, unsigned xm = 0u) {
  unsigned need = (unsigned)k;
  const unsigned maxb = xm ? ~uni(sh.minbits) : uni(sh.maxbits), diff = uni(sh.maxbits) ^ uni(sh.minbits);
  int remaining = diff ? 32 - __clz(diff) : 0;
  unsigned prefix = remaining < 32 ? (maxb >> remaining) : 0u;
  const int tid = tid_now();
  for (int i = tid; i < 2048; i += NT) hist[i] = 0;      // every pass leaves the histogram cleared (three barriers a pass)
  __syncthreads();
  while (remaining > 0) {
    const int w = remaining < 11 ? remaining : 11;
    const int shift = remaining - w;
    const unsigned dmask = (1u << w) - 1u;
    for (int p = 1 + tid; p <= n; p += NT) {
      const unsigned b = (unsigned)(H[p] >> 32) ^ xm;
      const unsigned hi = (shift + w < 32) ? (b >> (shift + w)) : 0u;
      if (hi == prefix) atomicAdd((unsigned *)&hist[(b >> shift) & dmask], 1u);
    }
    __syncthreads();
    {
      constexpr int BPT = 2048 / NT;                   // radix bins per thread
      static_assert(BPT * NT == 2048 && BPT >= 1, "the 2048 radix bins are scanned BPT per thread");
      unsigned h[BPT], pair = 0u;
#pragma unroll
      for (int x = 0; x < BPT; x++) { h[x] = hist[BPT * tid + x]; hist[BPT * tid + x] = 0u; pair += h[x]; }
      unsigned incl = pair;
      const int ln = tid & 63;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_down(incl, off, 64);
        if (ln + off < 64) incl += o;
      }
      if (ln == 0) sh.wsum[tid >> 6] = incl;
      __syncthreads();
      unsigned above = incl - pair;
      for (int wv = (tid >> 6) + 1; wv < NT / 64; wv++) above += sh.wsum[wv];
#pragma unroll
      for (int x = BPT - 1; x >= 0; x--) {
        if (above < need && need <= above + h[x]) { sh.sel_digit = (unsigned)(BPT * tid + x); sh.sel_need = need - above; sh.sel_count = h[x]; }
        above += h[x];
      }
    }
    __syncthreads();
    prefix = (prefix << w) | uni(sh.sel_digit);      // (written again only behind two more barriers)
    need = uni(sh.sel_need);
    remaining -= w;
  }
  return prefix;
}

#include "beam_sweep.h"

// ---- the events of the extraction loop, replayed by ONE wave (see the file header) -------------------
// Replays extraction i (1-based) when the tail position q = n - i + 1 may hold one of the top elements.
// Ranks 0..i-2 are out already, rank i-1 is at the root.  Wave 0 only; wave-uniform control flow.
//
// Who sits where: a heap position holds the best remaining element of its subtree that is not sitting
// further up.  Subtrees along a root-to-leaf chain are nested, so the occupants of a chain come out of ONE
// scan over the remaining ranks in order: the first element inside subtree(a_0) takes a_0, the next one
// inside subtree(a_1) takes a_1, and so on.
// A single wave runs this, so every dependent instruction costs its full latency: the scans keep the per-step
// work to a compare and a ballot (the chain scan reduces "inside subtree(a_d)" to "shares at least d path bits
// with q", computed once per element), fetch the next 64 ranks while the current ones are walked, and move
// values between lanes with v_readlane (the ballot's lane index is uniform), not with LDS permutes.
__device__ __forceinline__ bool scR_eq(const PruneMem &pm, int r, unsigned sc) { return ((const lds_u32 *)pm.compR)[2 * r + 1] == sc; }
constexpr int kMaxCand = 64;             // tail candidates replayed with the parallel scheme; more fall back to the serial loop
constexpr int kTakers = kMaxL + 2;       // occupants of one root-to-leaf chain

// Occupants of the chain root -> q (q = n - i + 1) after i - 1 extractions, found by one scan over the remaining
// ranks (see above).  Returns the rank of the element sitting AT q, or -1 when the chain ends earlier (the
// element has moved up, or out, before its tail turn).  takers[0..*ntake) receives the ranks of the occupants
// in chain order (the replay uses them to tell which later candidates an event can affect).  One wave.
// (Everything goes in and out by value: a reference or an out parameter of a function that is not inlined is a round
// trip through scratch memory.)  Returns (rank at q, or -1) << 32 | number of occupants written.
__device__ __noinline__ unsigned long long chain_scan(const lds_u32 *vposR, int nB, int n, int i, lds_i32 *takers) {
  vposR = uni(vposR);
  nB = uni(nB); n = uni(n); i = uni(i); takers = uni(takers);
  const int lane = threadIdx.x & 63;
  const unsigned q = (unsigned)(n - i + 1);
  const int Lq = 31 - __clz((int)q);
  const int nchunk = (nB + 63) >> 6;
  int occq = -1, d = 0;
  bool walking = true;
  int c = (i - 1) >> 6;
  unsigned vn = (c * 64 + lane < nB) ? vposR[c * 64 + lane] : 0u;
  for (; c < nchunk && walking; c++) {
    const int r = c * 64 + lane;
    const unsigned v = (r >= i - 1) ? vn : 0u;
    if (c + 1 < nchunk) vn = ((c + 1) * 64 + lane < nB) ? vposR[(c + 1) * 64 + lane] : 0u;
    // m = how many levels of the chain root -> q contain v: v is inside subtree(q >> (Lq - d)) iff d <= m
    int m = -1;
    if (v != 0u) {
      const int Lv = 31 - __clz((int)v);
      const int L = Lv < Lq ? Lv : Lq;
      const unsigned x = (v >> (Lv - L)) ^ (q >> (Lq - L));
      m = L - (x ? 32 - __clz((int)x) : 0);
    }
    int from = 0;
    for (;;) {
      const unsigned long long mk = __ballot(m >= d && lane >= from);
      if (!mk) break;
      const int l = __ffsll((long long)mk) - 1;
      if (lane == 0 && d < kTakers) takers[d] = c * 64 + l;
      if (d == Lq) { occq = c * 64 + l; d++; walking = false; break; }
      d++; from = l + 1;
    }
  }
  return ((unsigned long long)(unsigned)occq << 32) | (unsigned)(d < kTakers ? d : kTakers);
}

// The event itself: the element of rank rs leaves the tail position q = n - i + 1, the root element is output,
// and s runs down the path of larger children among the elements still in the heap (size n - i) until it is
// >= the larger child (:1372-1381).  The larger child of the hole (left on ties) is the best remaining element
// of the hole's subtree: the same kind of single scan.  Then s takes its place among the equal scores still in
// the heap (ranks >= i) by the pre-order of the positions.  Returns the new rank and the new position (packed, see the end).
__device__ __noinline__ unsigned long long apply_event(lds_u64 *compR_, lds_u32 *vposR_, lds_u32 *idR_, int nB, int n, int k, int i, int rs) {
  PruneMem pm;
  pm.compR = uni(compR_); pm.vposR = uni(vposR_); pm.idR = uni(idR_);
  nB = uni(nB); n = uni(n); k = uni(k); i = uni(i); rs = uni(rs);
  const lds_u32 *vposR = pm.vposR;
  const int lane = threadIdx.x & 63;
  const int nchunk = (nB + 63) >> 6;
  const unsigned ssc = (unsigned)(pm.compR[rs] >> 32);
  const unsigned hs = (unsigned)(n - i);
  unsigned hole = 1u;
  {
    bool walking = true;
    int Lh = 0;
    int c = i >> 6;
    const lds_u32 *scR = (const lds_u32 *)pm.compR;      // score bits = the high word of a composite
    unsigned vn = (c * 64 + lane < nB) ? vposR[c * 64 + lane] : 0u;
    unsigned sn = (c * 64 + lane < nB) ? scR[2 * (c * 64 + lane) + 1] : 0u;
    for (; c < nchunk && walking; c++) {
      const int r = c * 64 + lane;
      const unsigned v = (r >= i && r != rs) ? vn : 0u;
      const unsigned scv = sn;
      if (c + 1 < nchunk) {
        vn = ((c + 1) * 64 + lane < nB) ? vposR[(c + 1) * 64 + lane] : 0u;
        sn = ((c + 1) * 64 + lane < nB) ? scR[2 * ((c + 1) * 64 + lane) + 1] : 0u;
      }
      const int Lv = v ? 31 - __clz((int)v) : -1;
      int from = 0;
      for (;;) {
        if (2u * hole > hs) { walking = false; break; }
        // strictly below the hole, and the child of the hole on the way there is inside the heap
        const int dd = Lv - Lh;
        const bool below = dd > 0 && (v >> dd) == hole && (v >> (dd - 1)) <= hs;
        const unsigned long long mk = __ballot(below && lane >= from);
        if (!mk) break;
        const int l = __ffsll((long long)mk) - 1;
        if (ssc >= (unsigned)__builtin_amdgcn_readlane((int)scv, l)) { walking = false; break; }
        const unsigned vl = (unsigned)__builtin_amdgcn_readlane((int)v, l);
        hole = vl >> ((31 - __clz((int)vl)) - Lh - 1);
        Lh++;
        from = l + 1;
      }
    }
  }
  // the equal scores still in the heap, [g0, g1): one look at the 64 ranks around rs, loops only past its edges
  int g0, g1;
  {
    const int r = rs - 32 + lane;
    const bool eq = r >= i && r < nB && scR_eq(pm, r, ssc);
    const unsigned long long mk = __ballot(eq);
    const unsigned below = (unsigned)mk, above = (unsigned)(mk >> 33);     // ranks rs-32..rs-1 / rs+1..rs+31
    const int ndn = below == 0xffffffffu ? 32 : __clz((int)~below);
    const int nup = (above & 0x7fffffffu) == 0x7fffffffu ? 31 : __ffs((int)~above) - 1;
    g0 = rs - ndn; g1 = rs + 1 + nup;
    if (ndn == 32) while (g0 > i && scR_eq(pm, g0 - 1, ssc)) g0--;
    if (nup == 31) while (g1 < nB && scR_eq(pm, g1, ssc)) g1++;
  }
  const unsigned hk = prekey(hole);
  int cnt = 0;
  for (int base = g0; base < g1; base += 64) {
    const int r = base + lane;
    const bool before = r < g1 && r != rs && prekey(pm.vposR[r]) < hk;
    cnt += __popcll(__ballot(before));
  }
  const int newr = g0 + cnt;
  {
    // the ranks between the old and the new place move by one: 64 at a time, every lane reads before any lane writes
    const unsigned sid = pm.idR[rs];
    const unsigned long long sc = pm.compR[rs];
    if (newr < rs) {
      for (int top = rs; top > newr; top -= 64) {            // r-1 -> r for r in (newr, top], highest block first
        const int r = top - lane;
        const bool mv = r > newr;
        const unsigned a = mv ? pm.vposR[r - 1] : 0u, b = mv ? pm.idR[r - 1] : 0u;
        const unsigned long long cc = mv ? pm.compR[r - 1] : 0ull;
        wave_sync();
        if (mv) { pm.vposR[r] = a; pm.idR[r] = b; pm.compR[r] = cc; }
        wave_sync();
      }
    } else {
      for (int bot = rs; bot < newr; bot += 64) {            // r+1 -> r for r in [bot, newr), lowest block first
        const int r = bot + lane;
        const bool mv = r < newr;
        const unsigned a = mv ? pm.vposR[r + 1] : 0u, b = mv ? pm.idR[r + 1] : 0u;
        const unsigned long long cc = mv ? pm.compR[r + 1] : 0ull;
        wave_sync();
        if (mv) { pm.vposR[r] = a; pm.idR[r] = b; pm.compR[r] = cc; }
        wave_sync();
      }
    }
    if (lane == 0) { pm.vposR[newr] = hole; pm.idR[newr] = sid; pm.compR[newr] = sc; }
  }
  __builtin_amdgcn_wave_barrier();
  // new rank | bit 31: an equal score is still in the heap || new position
  return ((unsigned long long)hole << 32) | (unsigned)newr | (g1 - g0 > 1 ? 0x80000000u : 0u);
}

// one tail candidate handled start to finish by one wave (the serial form: more than kMaxCand candidates)
// Returns the turn at which the re-inserted element sits on a tail position again IF it is tied with an element still
// in the heap (the replay must then reach that turn), else 0.
__device__ __noinline__ int replay_tail(const PruneMem &pm, int nB, int n, int k, int i) {
  const int occ = uni((int)(chain_scan(pm.vposR, nB, n, i, pm.takers + kMaxCand * (kTakers + 1)) >> 32));
  if (occ < 0) return 0;
  const unsigned long long ev = apply_event(pm.compR, pm.vposR, pm.idR, nB, n, k, i, occ);
  const unsigned hole = uni((unsigned)(ev >> 32));
  const bool tied = (uni((unsigned)ev) & 0x80000000u) != 0u;
  int again = 0;
  if (hole >= (unsigned)(n - k + 1)) {                               // it sits on a tail position again: its turn comes later
    if ((threadIdx.x & 63) == 0) atomicOr((unsigned *)&pm.tailmask[(n - (int)hole) >> 5], 1u << ((n - (int)hole) & 31));
    if (tied) again = n - (int)hole + 1;
  }
  __builtin_amdgcn_wave_barrier();
  return again;
}

// sort_token_no_order() (:1492): the visiting order of the next frame.  keys[i] = score bits of token i in
// creation order.  Writes the token ids into svid[0..return value).  Whole workgroup.
//
// WIDE (the wide-beam layout): the heap is laid over the list areas -- it is dead once the top elements are
// collected, so they travel through `G` (a scratch array in the utterance's slice) with their token ids, and the
// sorted list is built where the heap was; vposR lies over the sorting scratch.
// FULL (the multipath frame's mid-frame sort, beam_exact_mp.h): the caller wants tindex[] WHOLE -- arr_full[0..n) = the token
// ids at array positions 0..n-1 after the sort, residual heap and extracted part -- beside svid[] (the part the next step
// visits).  Both directions then run sweep replay + sift replay (the closed form of the downward sort, mirrored for the
// upward one), or, where that cannot run, the extraction loop itself.
template <bool WIDE, int NT, bool FULL = false>
__device__ __forceinline__ int exact_prune(XShared &sh, const unsigned *keys, int n, int k, lds_u64 *H, int heap_cap,
                           unsigned long long *Hglob, PruneMem pm, lds_i32 *svid, int mode, u32x4 *G,
                           unsigned long long *tp = nullptr, int *arr_full = nullptr) {
  const int tid = tid_now();
  unsigned long long tc_ = tp ? wall_clock64() : 0ull, tc3_ = tc_;
  (void)tc3_;
#define PTICK(i) do { if (tp && tid == 0 && ((JAMD_XBEAM_PROBE != 3 && JAMD_XBEAM_PROBE != 5 && JAMD_XBEAM_PROBE != 6) || (i) == 7)) { const unsigned long long n_ = wall_clock64(); tp[i] += n_ - tc_; tc_ = n_; tc3_ = n_; } } while (0)
#ifdef JAMD_DEV
#define PTICK5(i) do { if (JAMD_XBEAM_PROBE == 5 && tp && tid == 0) { const unsigned long long n_ = wall_clock64(); tp[i] += n_ - tc3_; tc3_ = n_; } } while (0)
#define PTICK3(i) do { if (JAMD_XBEAM_PROBE == 3 && tp && tid == 0) { const unsigned long long n_ = wall_clock64(); tp[i] += n_ - tc3_; tc3_ = n_; } } while (0)
#define PTICK6(i) do { if (JAMD_XBEAM_PROBE == 6 && tp && tid == 0) { const unsigned long long n_ = wall_clock64(); tp[i] += n_ - tc3_; tc3_ = n_; } } while (0)
#else
#define PTICK5(i) ((void)0)
#define PTICK3(i) ((void)0)
#define PTICK6(i) ((void)0)
#endif
  if (n <= k) {
    for (int j = tid; j < n; j += NT) svid[j] = j;
    if constexpr (FULL) { for (int j = tid; j < n; j += NT) arr_full[j] = j; }
    __syncthreads();
    return n;
  }
#define PSTAT(i, v) do { if (pm.pstat && tid == 0) pm.pstat[i] += (v); } while (0)
  PSTAT(0, 1);
  const bool upward = k < n - k;
  const bool in_lds = n <= heap_cap;
  auto run = [&](auto Hh) -> void {
    constexpr bool kLdsHeap = std::is_same<decltype(Hh), lds_u64 *>::value;
    auto build_heap = [&]() {                                      // first loop of sort_token_upward / _downward (:1354-1367)
      for (int i = tid; i < n; i += NT) Hh[i + 1] = ((unsigned long long)keys[i] << 32) | (unsigned)i;
      __syncthreads();
      PTICK5(4);
      bool heaped = false;
      if constexpr (kLdsHeap) heaped = upward ? heapify_overlapped<true, NT>(Hh, n) : heapify_overlapped<false, NT>(Hh, n);
      if (!heaped) { if (upward) heapify_levels<true, NT>(Hh, n); else heapify_levels<false, NT>(Hh, n); }
    };
    build_heap();
    PTICK5(5);
    PTICK(4);
    bool done = false;
    // sort_token_downward() (beam < tokens <= 2 beam) has a closed form too: the same lists over the n - k SMALLEST
    // elements of the min-heap (score bits complemented), the sweep replay for their moves, and a replay of the short
    // sifts below the extracted region for the residual heap (down_finish()).  Full shape with the sweep's scratch only.
    bool down_ok = false;
    // (wide layout, full shape only: the narrow layout's beams have a handful of tail candidates, and the mere presence of
    // this code in the kernel costs its steps 0-C 2.5 us each per frame at beam 800 -- profiles/r04_ab_sweep_code_presence.txt)
    // (round 5: the whole-array form also in the half shape -- the multipath frame's mid-frame sort needs it there; what does
    // not fit half a CU's LDS falls back to the extraction loop at run time)
    if constexpr (kLdsHeap && (WIDE || FULL) && (NT == jamdb::NT || FULL)) down_ok = (FULL || !upward) && pm.sw_glob != nullptr;
    // (FULL: down_ok = "the whole array can come out of the closed form", either direction)
    const int cnt = upward ? k : n - k;                            // extractions
    const unsigned xm = upward ? 0u : 0xffffffffu;
    if ((upward || down_ok) && (!FULL || down_ok) && mode != 1 && pm.b_cap > 0) {
      // closed form of the extraction loop
      const unsigned vk = kth_largest<NT>(sh, Hh, n, cnt, pm.hist, xm);
      // The top list sorted by (score descending, pre-order of the heap position ascending).  A bitonic network is 55
      // dependent steps at this size; the scores are spread well over their range, so the list is sorted by counting
      // instead: 2048 score bins between the k-th largest score and the maximum (monotone in the score, equal scores
      // in one bin), a prefix sum over the bins, a scatter by bin, and inside its bin (a handful of entries unless many
      // scores are equal) every entry counts the composites greater than its own.  The composites are distinct.
      const unsigned span = (upward ? uni(sh.maxbits) : ~uni(sh.minbits)) - vk;
      const int bshift = span ? max(0, 32 - __clz(span) - 11) : 0;
      // Bins over [k-th score, best]: linear in the score bits -- or, when the list crowds a few linear bins (peaked
      // scores: most of the beam sits just above the cut, a few tokens far above), on a LOG scale of the distance from
      // the cut: 32 octaves x 64 steps, fine where the list is dense.  Both are monotone in the score.
      bool logbins = false;
      auto bin_of = [&](unsigned scb) {
        const unsigned dlt = scb - vk;
        if (!logbins) return (int)min(2047u, dlt >> bshift);
        if (dlt == 0u) return 0;
        const int lz = __clz((int)dlt);
        return (int)((unsigned)(31 - lz) << 6 | ((lz == 31 ? 0u : (dlt << (lz + 1))) >> 26));
      };
      if (tid == 0) { sh.nB = 0; sh.i_last = 0; sh.sel_count = 0u; }
      for (int i = tid; i < (cnt + 31) / 32 + 1; i += NT) pm.tailmask[i] = 0u;
      __syncthreads();                                   // (kth_largest() left the histogram cleared)
      for (int p0 = 1; p0 <= n; p0 += NT) {
        const int p = p0 + tid;
        const unsigned long long hv = p <= n ? Hh[p] : 0ull;
        const unsigned hi = (unsigned)(hv >> 32) ^ xm;
        if ((FULL || !upward) && p <= n) Hglob[p] = hv;  // the heap itself: down_finish() replays the sifts below the extracted region on it
        const bool in = p <= n && hi >= vk;
        const int slot = wave_alloc(&sh.nB, in);
        if (in && slot < pm.b_cap) {
          const unsigned pk = 0xffffffffu - prekey((unsigned)p);
          if constexpr (WIDE) G[slot] = u32x4{pk, hi, (unsigned)hv, 0u};
          else pm.compT[slot] = ((unsigned long long)hi << 32) | pk;
          atomicAdd((unsigned *)&pm.hist[bin_of(hi)], 1u);
        }
      }
      __syncthreads();
      const int nB = uni(sh.nB);
      if (nB <= pm.b_cap) {
        // crowded bins (an entry counts the larger composites of its bin: quadratic in the bin) -> count again on the log scale
        constexpr int BPT0 = 2048 / NT;
        unsigned mx = 0u;
#pragma unroll
        for (int x = 0; x < BPT0; x++) mx = max(mx, pm.hist[BPT0 * tid + x]);
        if (mx > 48u) atomicMax(&sh.sel_count, mx);        // (cleared with nB above)
        __syncthreads();
        if (uni(sh.sel_count) > 48u) {
          logbins = true;
          for (int i = tid; i < 2048; i += NT) pm.hist[i] = 0u;
          __syncthreads();
          for (int e = tid; e < nB; e += NT) {
            unsigned hi;
            if constexpr (WIDE) hi = G[e].y; else hi = (unsigned)(pm.compT[e] >> 32);
            atomicAdd((unsigned *)&pm.hist[bin_of(hi)], 1u);
          }
          __syncthreads();
        }
      }
      PTICK(5);
      if (nB <= pm.b_cap) {
        {
          // exclusive prefix from the top bin down: thread t owns bins 2047 - BPT t ... 2048 - BPT (t + 1)
          constexpr int BPT = 2048 / NT;
          unsigned cb[BPT]; int tot = 0;
#pragma unroll
          for (int x = 0; x < BPT; x++) { cb[x] = pm.hist[2047 - BPT * tid - x]; tot += (int)cb[x]; }
          unsigned ex = (unsigned)block_excl_scan<NT>(sh, tot);
#pragma unroll
          for (int x = 0; x < BPT; x++) { pm.hist[2047 - BPT * tid - x] = ex; ex += cb[x]; }
          __syncthreads();
          for (int e = tid; e < nB; e += NT) {                  // by bin, any order inside; hist[b] ends as the END of bin b
            if constexpr (WIDE) {                               // (the heap is dead: every thread is past the barrier behind the collection)
              const u32x4 g = G[e];
              const unsigned at = atomicAdd((unsigned *)&pm.hist[bin_of(g.y)], 1u);
              pm.compR[at] = ((unsigned long long)g.y << 32) | g.x;
              pm.idT[at] = g.z;
            } else {
              const unsigned long long c = pm.compT[e];
              pm.compR[atomicAdd((unsigned *)&pm.hist[bin_of((unsigned)(c >> 32))], 1u)] = c;
            }
          }
          __syncthreads();
          for (int e = tid; e < nB; e += NT) {
            const unsigned long long c = pm.compR[e];
            const int b = bin_of((unsigned)(c >> 32));
            const int lo = b == 2047 ? 0 : (int)pm.hist[b + 1], hi = (int)pm.hist[b];
            int r = lo;
            for (int x = lo; x < hi; x++) r += pm.compR[x] > c ? 1 : 0;
            pm.compT[r] = c;
            if constexpr (WIDE) pm.idR[r] = pm.idT[e];
          }
          __syncthreads();
        }
        { lds_u64 *t_ = pm.compR; pm.compR = pm.compT; pm.compT = t_; }     // the sorted list is what the replay calls compR
        if constexpr (WIDE) pm.vposR = (lds_u32 *)pm.compT;                 // (the sorting scratch is free from here on)
        // An event re-inserts ONE element; the other elements keep their places in the order, and an element whose
        // score is unique in the list is ranked by its score wherever it sits.  So the replay can stop behind the last
        // turn at which a TIED element may sit on the tail position (sh.i_last; such an element is one that starts on a
        // tail position, possibly re-inserted on a later one): the events after it cannot change the order.
        for (int r = tid; r < nB; r += NT) {
          const unsigned long long cr = pm.compR[r];
          const unsigned p = prekey_pos(0xffffffffu - (unsigned)cr);
          pm.vposR[r] = p;
          if constexpr (!WIDE) pm.idR[r] = (unsigned)Hh[p];
          if (p >= (unsigned)(n - cnt + 1)) {
            atomicOr((unsigned *)&pm.tailmask[(n - (int)p) >> 5], 1u << ((n - (int)p) & 31));
            const unsigned sc = (unsigned)(cr >> 32);
            const bool tied = (r > 0 && (unsigned)(pm.compR[r - 1] >> 32) == sc) || (r + 1 < nB && (unsigned)(pm.compR[r + 1] >> 32) == sc);
            if (tied) atomicMax(&sh.i_last, n - (int)p + 1);
          }
        }
        __syncthreads();
        PTICK(6);
        if (FULL || !upward) {
          // the residual heap: every event matters (the survivors stay in heap layout), so the sweep runs over all turns
          bool ok = false;
          if constexpr (kLdsHeap && (WIDE || FULL) && (NT == jamdb::NT || FULL)) {
            const int tailb = sweep_down_bytes(cnt);
            // (the sift replay holds the whole heap in LDS, 8 bytes a token: a frame too large for it goes to the extraction loop at once)
            if (pm.sw_bytes > tailb + 1024 && 8 * (n + 2) + cnt + 80 <= ((pm.sw_bytes - tailb) & ~15) && n < 0xffff) {
              SweepDown dn;
              unsigned char JAMD_LDS *tl = pm.sw_region + ((pm.sw_bytes - tailb) & ~15);
              dn.fd = (lds_u32 *)tl; dn.posend = dn.fd + cnt + 1; dn.evbits = dn.posend + kSwLeft;
              dn.want_order = FULL ? 1 : 0;
              const int evmax = sweep_pick_evmax(nB, cnt, (pm.sw_bytes - tailb) & ~15);
              if (evmax) ok = sweep_replay<NT>(sh, pm.sw_region, (pm.sw_bytes - tailb) & ~15, pm.sw_glob, pm.compR, pm.vposR, pm.idR, pm.tailmask, nB, n, cnt,
                                               cnt, svid, evmax, &dn);
              __syncthreads();
              if constexpr (FULL) {                          // svid[] = the extracted elements, last extracted first: tindex[n - cnt ..)
                if (ok) for (int j = tid; j < cnt; j += NT) arr_full[n - cnt + j] = svid[j];
                __syncthreads();
              }
              if (ok) ok = upward ? down_finish<NT, false>(sh, pm.sw_region, (pm.sw_bytes - tailb) & ~15, Hglob, n, n - cnt, dn, sweep_ids(pm.sw_glob), nB, svid, FULL ? arr_full : nullptr)
                                  : down_finish<NT, true>(sh, pm.sw_region, (pm.sw_bytes - tailb) & ~15, Hglob, n, n - cnt, dn, sweep_ids(pm.sw_glob), nB, svid, FULL ? arr_full : nullptr);
              if constexpr (FULL) {                          // downward: what the next step visits is the residual heap
                if (ok && !upward) { __threadfence_block(); __syncthreads(); for (int j = tid; j < k; j += NT) svid[j] = arr_full[j]; }
              }
              if (!ok && tid == 0) sh.sw_info = -1;
              __syncthreads();
            }
          }
          if (ok) { done = true; PTICK(7); PSTAT(5, 1); PSTAT(7, sh.sw_info); }
          else build_heap();
        } else {
        // Tail positions holding a top element, in the order of their turns (bit b <-> extraction b + 1).  The chain
        // scans only READ the rank lists, so all candidates are scanned at once, one wave each; wave 0 then walks
        // the candidates in turn order and applies the events.  An event moves one element (and shifts the ranks
        // inside its tie group): a later candidate is scanned again only if that can change its chain.
        lds_i32 *ctl = pm.need + kMaxCand;                    // ncand, cursor, finished, last turn that matters
        if (tid < 64) {
          const int nw = (k + 31) / 32, lim = k;              // all of them: the last turn that matters can move back
          int nc = 0;
          for (int w0 = 0; w0 < nw; w0 += 64) {                // one mask word per lane
            const int w = w0 + tid, rem = lim - w * 32;        // turns i = w * 32 + b + 1 <= lim
            unsigned bits = w < nw ? pm.tailmask[w] : 0u;
            bits &= rem >= 32 ? 0xffffffffu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
            const int c = __popc(bits);
            int incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (tid >= off) incl += o; }
            int at = nc + incl - c;
            while (bits) {
              const int b = __ffs((int)bits) - 1;
              bits &= bits - 1u;
              if (at < kMaxCand) { pm.cand[at] = w * 32 + b + 1; pm.need[at] = 1; pm.ordv[at] = at; }
              at++;
            }
            nc += __shfl(incl, 63, 64);
          }
          if (tid == 0) { ctl[0] = nc; ctl[1] = 0; ctl[2] = 0; ctl[3] = sh.i_last; }
        }
        __syncthreads();
        // More live candidates (turn <= the last turn that matters) than the parallel replay holds: an event costs a
        // scan of the whole top list on one wave, hundreds of them cost more than the extraction loop itself run
        // pipelined -- give the closed form up for this frame.  (Wide beams over flat scores: a third of the top
        // elements sit on tail positions.)  The heap was overlaid by the lists in the wide layout: it is built again.
        // More candidates than the wave-serial replay below holds (wide beams: a tenth of the top elements sit on tail
        // positions, hundreds of candidates): the sweep replay resolves them together (beam_sweep.h).  What it cannot
        // hold -- or a work area without its scratch -- goes to the extraction loop itself: the heap is built again (the
        // sweep lays its own image over the whole overlay).
        bool give_up = false;
        if (uni(ctl[0]) > kMaxCand && uni(ctl[3]) > 0) {
          bool swept = false;
          if constexpr (WIDE && NT == jamdb::NT) {         // (the narrow layout and the half shape serve narrow beams: a handful of candidates)
           if (pm.sw_glob) {
            const int evmax = sweep_pick_evmax(nB, k, pm.sw_bytes);
            if (evmax) swept = sweep_replay<NT>(sh, pm.sw_region, pm.sw_bytes, pm.sw_glob, pm.compR, pm.vposR, pm.idR, pm.tailmask, nB, n, k,
                                                uni(ctl[3]), svid, evmax, nullptr);
            if (!swept && tid == 0) sh.sw_info = -1;
            __syncthreads();
           }
          }
          if (swept) { done = true; PTICK(7); PSTAT(3, 1); PSTAT(7, sh.sw_info); }
          else { give_up = true; PSTAT(4, 1); }
        }
        if (give_up) {
          build_heap();
        } else if (!done) {
#ifdef JAMD_DEV
        if (JAMD_XBEAM_PROBE == 3 && tp && tid == 0) tc3_ = wall_clock64();   // slot 7 - (4 + 5 + 6) = everything before the replay
#endif
        const int ncand0 = uni(ctl[0]);
        if (ncand0 > kMaxCand) {
          if (tid < 64) {                                     // serial form, straight off the mask, up to the last turn that matters
            const int nw = (k + 31) / 32;
            int ilast = uni(ctl[3]);
            for (int w = 0; w < nw && w * 32 + 1 <= ilast; w++) {
              unsigned donebits = 0u;
              for (;;) {
                const unsigned bits = ((volatile lds_u32 *)pm.tailmask)[w] & ~donebits;
                if (!bits) break;
                const int b = __ffs((int)bits) - 1;
                donebits |= (b == 31) ? 0xffffffffu : ((2u << b) - 1u);
                const int i = w * 32 + b + 1;
                if (i > ilast) break;
                if (i <= k) { const int again = uni(replay_tail(pm, nB, n, k, i)); if (again > ilast) ilast = again; }
              }
            }
          }
        } else if (ncand0 > 0 && uni(ctl[3]) > 0) {
          // A candidate keeps its slot (turn, occupant, chain occupants); ordv[] lists the slots in turn order, so a
          // candidate born of an event is one shifted int per later candidate.
          for (;;) {
            // (re)scan: the candidate at position c on wave c % 16
            {
              const int ncand = uni(ctl[0]), wv = uni(tid >> 6), ilast = uni(ctl[3]);
              for (int c = uni(ctl[1]) + wv; c < ncand; c += NT / 64) {
                const int sl = uni(pm.ordv[c]);
                const int turn = uni(pm.cand[sl]);
                if (!uni(pm.need[sl]) || turn > ilast) continue;
                lds_i32 *tk = pm.takers + sl * (kTakers + 1);
                const unsigned long long cs = chain_scan(pm.vposR, nB, n, turn, tk);
                if ((tid & 63) == 0) { pm.occ[sl] = (int)(cs >> 32); tk[kTakers] = (int)(unsigned)cs; pm.need[sl] = 0; }
              }
            }
            __syncthreads();
            PTICK3(4);
            if (tid < 64) {
              const int lane = tid;
              int ncand = uni(ctl[0]), c = uni(ctl[1]), ilast = uni(ctl[3]);
              for (; c < ncand; c++) {
                const int sl = uni(pm.ordv[c]);
                const int i = uni(pm.cand[sl]), nd = uni(pm.need[sl]), rs = uni(pm.occ[sl]);
                if (i > ilast) { c = ncand; break; }          // nothing behind this turn can change the order
                if (nd) break;                                // invalidated by an earlier event: next round
                if (rs < 0) continue;
                PTICK6(4);
                const unsigned long long ev = apply_event(pm.compR, pm.vposR, pm.idR, nB, n, k, i, rs);
                PTICK6(5);
                const unsigned hole = uni((unsigned)(ev >> 32));
                const int newr = uni((int)((unsigned)ev & 0x7fffffffu));
                const bool tied = (uni((unsigned)ev) & 0x80000000u) != 0u;
#ifdef JAMD_DEV
                if ((JAMD_XBEAM_PROBE == 3 || JAMD_XBEAM_PROBE == 6) && tp && tid == 0) tp[6] += 100;   // events (1 us each)
#endif
                const int lo = newr < rs ? newr : rs, hi = newr < rs ? rs : newr;
                // which later candidates can this change?  (ranks outside [lo, hi] keep their numbers)  One lane each;
                // the chain occupants are read with a fixed trip count so the loads go out together.
                const int c2 = c + 1 + lane;                  // ncand <= kMaxCand = 64: one pass
                int sl2 = 0, i2 = 0x7fffffff;
                if (c2 < ncand) { sl2 = pm.ordv[c2]; i2 = pm.cand[sl2]; }
                if (c2 < ncand && !pm.need[sl2] && hi >= i2 - 1) {        // (hi < i2 - 1: s is out before that turn)
                  const lds_i32 *tk = pm.takers + sl2 * (kTakers + 1);
                  const int nt = tk[kTakers], oc2 = pm.occ[sl2];
                  bool hit = false; int dat = 0;
#pragma unroll
                  for (int x = 0; x < kTakers; x++) {
                    const int tr = tk[x];
                    if (x < nt) { if (tr >= lo && tr <= hi) hit = true; if (tr < lo) dat++; }
                  }
                  if (!hit && !(oc2 >= 0 && oc2 < lo)) {
                    // would s, now at `hole`, be taken when the walk passes it?  It shares m levels with the chain.
                    const unsigned q2 = (unsigned)(n - i2 + 1);
                    const int Lq2 = 31 - __clz((int)q2), Lv = 31 - __clz((int)hole);
                    const int L = Lv < Lq2 ? Lv : Lq2;
                    const unsigned x = (hole >> (Lv - L)) ^ (q2 >> (Lq2 - L));
                    const int m = L - (x ? 32 - __clz((int)x) : 0);
                    hit = m >= dat;
                  }
                  if (hit) pm.need[sl2] = 1;
                }
                wave_sync();
                if (hole >= (unsigned)(n - k + 1)) {          // s sits on a tail position again: a candidate with a later turn
                  const int inew = n - (int)hole + 1;
                  if (tied && inew > ilast) ilast = inew;
                  const unsigned long long eqm = __ballot(c2 < ncand && i2 == inew);
                  if (eqm) {                                  // already a candidate: two elements share the position now
                    const int e = c + 1 + (__ffsll((long long)eqm) - 1);
                    if (lane == 0) pm.need[pm.ordv[e]] = 1;
                  } else {
                    if (ncand >= kMaxCand) { if (lane == 0) ctl[0] = kMaxCand + 1; ncand = kMaxCand + 1; break; }   // overflow: finish serially
                    const int at = c + 1 + __popcll(__ballot(c2 < ncand && i2 < inew));
                    const int mvslot = (c2 >= at && c2 < ncand) ? sl2 : -1;
                    wave_sync();
                    if (mvslot >= 0) pm.ordv[c2 + 1] = mvslot;
                    if (lane == 0) { pm.ordv[at] = ncand; pm.cand[ncand] = inew; pm.need[ncand] = 1; pm.occ[ncand] = -1; ctl[0] = ncand + 1; }
                    ncand++;
                  }
                  wave_sync();
                }
              }
              if (lane == 0) { ctl[1] = c; ctl[2] = (c >= ncand || ncand > kMaxCand) ? 1 : 0; ctl[3] = ilast; }
            }
            __syncthreads();
            PTICK3(5);
            if (uni(ctl[2])) break;
          }
          if (ctl[0] > kMaxCand && tid < 64) {                // candidate table overflowed mid-way: the rest serially
            int ilast = uni(ctl[3]);
            for (int i = pm.cand[pm.ordv[ctl[1]]]; i <= k && i <= ilast; i++) {
              bool any = false;
              for (int r0 = 0; r0 < nB; r0 += 64) { const int r = r0 + (tid & 63); if (__ballot(r < nB && pm.vposR[r] == (unsigned)(n - i + 1))) { any = true; break; } }
              if (any) { const int again = uni(replay_tail(pm, nB, n, k, i)); if (again > ilast) ilast = again; }
            }
          }
        }
        __syncthreads();
        PTICK(7);
        PSTAT(uni(ctl[3]) > 0 && uni(ctl[0]) > 0 ? 2 : 1, 1);
        for (int j = tid; j < k; j += NT) svid[j] = (int)pm.idR[k - 1 - j];    // tindex[n-k+j]: ascending
        done = true;
        }                                                        // (!give_up)
        }                                                        // (upward)
      }                                                          // (more ties on the cut than the lists hold: the heap is untouched)
    }
    if (!done) {
      PSTAT(6, 1);
      // the extraction loop itself: pipelined on one wave when the heap is in LDS, else (and in the cross-check
      // mode JAMD_ORDER_EXACT_SERIAL) sequentially on one lane
      bool piped = false;
      if constexpr (std::is_same<decltype(Hh), lds_u64 *>::value) {
        if (mode != 1) {
          if (tid < 64) { if (upward) heap_extract_pipelined<true>(Hh, n, k); else heap_extract_pipelined<false>(Hh, n, n - k); }
          piped = true;
        }
      }
      if (!piped && tid == 0) { if (upward) heap_extract_serial<true>(Hh, n, k); else heap_extract_serial<false>(Hh, n, n - k); }
      __syncthreads();
      for (int j = tid; j < k; j += NT) svid[j] = (int)(unsigned)(upward ? Hh[n - k + 1 + j] : Hh[1 + j]);
      if constexpr (FULL) { for (int p = tid; p < n; p += NT) arr_full[p] = (int)(unsigned)Hh[p + 1]; }
    }
    __syncthreads();
  };
  if (in_lds) run(H); else run(Hglob);
  return k;
#undef PSTAT
#undef PTICK
}

// outprob_cd() with IWCD_NBEST (outprob.c:330-365): the mean of the K best member scores of a state set, `lps` lanes
// per set (a power of two, the lanes of a set adjacent).  Each lane keeps the K best of its members in descending
// order (insertion by max / min), the lanes merge in a butterfly; the sum runs from the best down as in the reference.
template <int K>
__device__ __forceinline__ float nbest_of_set(const LexDev &lx, const XRowRef &row, int a, int bnd, int sub, int lps) {
  float b[K];
#pragma unroll
  for (int i = 0; i < K; i++) b[i] = JAMD_LOG_ZERO;
  int n = 0;
  auto ins = [&](float p) {
#pragma unroll
    for (int i = 0; i < K; i++) { const float hi = __builtin_fmaxf(b[i], p); p = __builtin_fminf(b[i], p); b[i] = hi; }
  };
  for (int m = a + sub; m < bnd; m += 8 * lps) {
    int ix[8]; float pv[8];
#pragma unroll
    for (int jj = 0; jj < 8; jj++) ix[jj] = (m + lps * jj < bnd) ? lx.set_states(m + lps * jj) : -1;
#pragma unroll
    for (int jj = 0; jj < 8; jj++) pv[jj] = (ix[jj] >= 0) ? row[ix[jj]] : JAMD_LOG_ZERO;
#pragma unroll
    for (int jj = 0; jj < 8; jj++) { n += pv[jj] > JAMD_LOG_ZERO ? 1 : 0; ins(pv[jj]); }
  }
  for (int off = 1; off < lps; off <<= 1) {
    float c[K];
#pragma unroll
    for (int i = 0; i < K; i++) c[i] = __shfl_xor(b[i], off, 64);
    n += __shfl_xor(n, off, 64);
#pragma unroll
    for (int i = 0; i < K; i++) ins(c[i]);
  }
  if (n > lx.cdmax_num) n = lx.cdmax_num;
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < K; i++) if (n > i) sum += b[i];
  return sum / (float)n;
}

// The survivors of the frame, in visiting order: in LDS (narrow layout) or in the utterance's slice (wide layout:
// steps 0 and A read them in order, only the winner look-ups of step C are gathers -- from L2).
template <bool WIDE> struct XSv;
template <> struct XSv<false> {
  lds_v4 *p;
  __device__ __forceinline__ Tok load(int j) const { return lds_tok_load(p, j); }
  __device__ __forceinline__ void store(int j, const Tok &t) const { lds_tok_store(p, j, t); }
  __device__ __forceinline__ void quads(int j, u32x4 &a, u32x4 &b) const { a = p[2 * j]; b = p[2 * j + 1]; }
};
template <> struct XSv<true> {
  u32x4 *p;
  __device__ __forceinline__ void quads(int j, u32x4 &a, u32x4 &b) const { a = p[2 * j]; b = p[2 * j + 1]; }
  __device__ __forceinline__ Tok load(int j) const {
    u32x4 a, b;
    quads(j, a, b);
    Tok t;
    t.node = (int)a.x; t.score = __uint_as_float(a.y); t.last_tre = (int)a.z; t.last_cword = (int)a.w;
    t.last_lscore = __uint_as_float(b.x); t.last_wid = (int)b.y; t.pad0 = (int)b.z; t.pad1 = (int)b.w;
    return t;
  }
  __device__ __forceinline__ void store(int j, const Tok &t) const {
    p[2 * j] = u32x4{(unsigned)t.node, __float_as_uint(t.score), (unsigned)t.last_tre, (unsigned)t.last_cword};
    p[2 * j + 1] = u32x4{__float_as_uint(t.last_lscore), (unsigned)t.last_wid, (unsigned)t.pad0, (unsigned)t.pad1};
  }
};

// ---- the kernel's arguments, read where they are used ------------------------------------------------------------
// LexDev + XWork are some 150 dwords of launch constants, and the frame loop derives another forty uniform addresses
// from them.  Taken as by-value parameters they are all loaded at the kernel's entry and stay live across the frame
// loop: the hardware has ~100 SGPRs, so the compiler parked a thousand of them in VGPR lanes (sgpr_spill_count 1 047 in
// round 4) and a sixth of the instruction stream was v_readlane / v_writelane.  They are constants of the KERNARG
// segment, which a wave can read at any time with a scalar load (scalar data cache): the first two parameters are one
// struct at offset 0 of that segment, and every frame re-derives its view of it from an address the compiler cannot see
// through (xargs_now(): the same device as tid_now() for the thread index), so that a value is loaded in the phase that
// uses it and dies there.  JAMD_XARGS_RELOAD=0 builds the round-4 form (everything live from the kernel's entry).
#ifndef JAMD_XARGS_RELOAD
#define JAMD_XARGS_RELOAD 1
#endif
struct XKArgs { LexDev lx; XWork xw; };
__device__ __forceinline__ const XKArgs &xargs_now() {
  unsigned long long a = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  return *(const XKArgs *)(const __attribute__((address_space(4))) XKArgs *)a;    // constant address space: scalar loads
}

// everything the frame loop derives from the launch constants (declares lx, xw, wk and the LDS / slice views)
#define XBEAM_VIEWS(KA)                                                                                              \
  const LexDev &lx = (KA).lx; const XWork &xw = (KA).xw; const Work &wk = xw.w;                                        \
  XSv<WIDE> sv;                                            /* Tok[beam], two quads each */                           \
  if constexpr (WIDE) sv.p = reinterpret_cast<u32x4 *>(ub + wk.o_sv); else sv.p = (lds_v4 *)dyn_lds;                    \
  lds_i32 *sv_atom = (lds_i32 *)(dyn_lds + xw.off_atom);                                                               \
  lds_i32 *welist = (lds_i32 *)(dyn_lds + xw.off_we);      /* word ends of the frame; the pruning step returns its order here */ \
  lds_i32 *dbase = (lds_i32 *)(dyn_lds + xw.off_dbase);    /* [beam + 2] first dense visiting index of each source */ \
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
  pm.pstat = xw.o_sweep ? sh.pst : nullptr;               /* (a generic pointer to LDS: a handful of accesses per frame) */ \
  lds_u64 *Hlds = (lds_u64 *)(dyn_lds + xw.off_heap);                                                                  \
  unsigned long long *Hglob = reinterpret_cast<unsigned long long *>(ub + xw.o_heap);                                  \
  u32x4 *Gcol = reinterpret_cast<u32x4 *>(ub + xw.o_collect);                                                          \
  const float lmw = lx.lm_weight, pen = lx.lm_penalty;                                                                 \
  const bool dfa = lx.lm_type != JAMD_LM_NGRAM;                                                                        \
  const bool wordmode = lx.lm_type == JAMD_LM_WORD;                                                                    \
  unsigned long long *memo = reinterpret_cast<unsigned long long *>(ub + wk.o_lmcache);                                \
  const int s1 = xw.s1, XW = xw.xw;                                                                                    \
  const unsigned submask = (1u << s1) - 1u;                                                                            \
  const int nroot_x = wordmode ? 0 : (dfa ? lx.startnum : lx.isolatenum);                                              \
  (void)sv_atom; (void)welist; (void)dbase; (void)tpre; (void)rowc; (void)Hlds; (void)Hglob; (void)Gcol; (void)lmw; (void)pen; \
  (void)memo; (void)XW; (void)submask; (void)nroot_x

// JAMD_HALF_WAVES (development, tools/build_variant.sh): waves per SIMD the HALF shape is compiled for.  4 = 128 VGPRs (two
// workgroups fill a CU's register file); 5 = 96 VGPRs, which leaves a fifth of the file to a co-resident scoring wave
// (with JAMD_HALF_LDS_KB=62 also the LDS for one gmm_tile workgroup): the experiment of HISTORY.md part II section 5, "K1 beside K6x".
#ifndef JAMD_HALF_WAVES
#define JAMD_HALF_WAVES 4
#endif
template <bool TIMED, bool WIDE, int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == kHalfNT ? JAMD_HALF_WAVES : 4, NT == kHalfNT ? JAMD_HALF_WAVES : 4)))
beam_exact_kernel(XKArgs ka_, const float *__restrict__ scores, int S, const int *__restrict__ utt_off, int smode) {
  __shared__ XShared sh;
  extern __shared__ __align__(16) unsigned char dyn_lds[];
#if JAMD_XARGS_RELOAD
  const XKArgs &ka0 = xargs_now();
#else
  const XKArgs &ka0 = ka_;
#endif
  if (threadIdx.x == 0 && ka0.xw.w.resident) __hip_atomic_fetch_add(ka0.xw.w.resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // this workgroup holds its share of a CU now
  const int u = min(max(utt_off[gridDim.x + 1 + blockIdx.x], 0), (int)gridDim.x - 1);   // longest utterance first (upload_utt_off()); clamped: never outside the launch's slices
  int tid = threadIdx.x;                                                     // refreshed every frame: see tid_now()
  const int t_begin = utt_off[u], nrows = utt_off[u + 1] - t_begin;
  StreamState *ss = smode ? ka0.xw.w.stream + u : nullptr;
  const bool resume = smode && ss->started;
  const int base = resume ? ss->frames_done : 0;
  const int T = base + nrows;
  const bool finish = smode != 1;
  unsigned char *const ub = ka0.xw.w.slices + (size_t)u * ka0.xw.w.utt_stride;
#define SLICE(T, off, i) (*reinterpret_cast<T *>(ub + (unsigned)((off) + (unsigned)sizeof(T) * (unsigned)(i))))
#define NODEKEY(i) SLICE(unsigned long long, wk.o_nodekey, i)
#define NODEFIRST(i) SLICE(unsigned, xw.o_nodefirst, i)
#define CUR(i) SLICE(Tok, wk.o_cur, i)
#define CURKEY(i) SLICE(unsigned, wk.o_cur_key, i)
#define REC(i) SLICE(u32x4, wk.o_cur + (unsigned)sizeof(Tok) * (unsigned)wk.tok_cap, i)   /* {node, visiting index of the winner, trellis word or -2, score bits} */
#define TOUCHED(i) SLICE(int2, wk.o_touched, i)
#define ARCQ(i) SLICE(int2, wk.o_arcq, i)
#define ATOM(i) SLICE(jamd_trellis_atom, wk.o_atoms, i)
  jamd_pass1_result *res = ka0.xw.w.res + u;
  // LDS image: survivors in VISITING ORDER (no node hash: a candidate names its source by position)
  XBEAM_VIEWS(ka0);
  int *const pstat_glob = xw.o_sweep ? reinterpret_cast<int *>(ub + xw.o_pstat) : nullptr;
  if (tid == 0) for (int i = 0; i < 16; i++) sh.pst[i] = 0;
  for (int i = tid; i < cl.nslot; i += NT) { cl.lkey[i] = 0ull; cl.lnode[i] = -1; cl.lfirst[i] = 0u; }

  if (resume) {
    if (!ss->active) return;
    if constexpr (!WIDE) {
      const u32x4 *src = (const u32x4 *)(ub + wk.o_sv);
      for (int i = tid; i < wk.sv_bytes / 16; i += NT) sv.p[i] = src[i];
    }
    if (tid == 0) { sh.n_atom = ss->n_atom; sh.n_surv = ss->n_surv; }
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
    __syncthreads();
    if (nrows <= 0) {
      if (tid == 0) { if (smode != 1) res->status = JAMD_PASS1_FAIL; if (ss) { ss->started = 0; ss->active = 1; } }
      return;
    }
    // get_back_trellis_init(): the silB head token (init_nodescore, beam.c:1622-1665); grammar / word list:
    // the initial tokens enter through the finalize and pruning steps of a pseudo frame 0
    if (tid == 0 && !dfa) {
      const int node = lx.word_head(lx.head_silwid);
      const int4 nr = lx.node_b(node);
      Tok nw;
      float ls = (nr.y != 0) ? max_successor_prob(lx, -1, nr.y) : 0.0f;
      ls = ls * lmw + pen;
      nw.node = node; nw.last_tre = -1; nw.last_cword = -1; nw.last_wid = -1; nw.last_lscore = ls;
      nw.score = node_outprob(lx, scores + (size_t)t_begin * S, nr.w, nr.z, -1) + ls;
      nw.pad0 = nr.x; nw.pad1 = 0;
      sv.store(0, nw);
      sh.n_surv = 1;
    }
  }
  float thr = resume ? ss->thr : JAMD_LOG_ZERO;
  // the phase clocks of the instrumented instantiation live in LDS (thread 0 adds to them): eight 64-bit counters in
  // registers cost the kernel 16 VGPRs it does not have
  unsigned long long *const ph = sh.ph;
  if (TIMED && threadIdx.x == 0) for (int i = 0; i < 8; i++) sh.ph[i] = 0ull;
  unsigned long long tc = wall_clock64(), tc2 = tc;
  (void)tc2;
#ifdef JAMD_DEV
  const unsigned long long cyc0 = clock64(), wall0 = tc;   // JAMD_XBEAM_PROBE == 4: shader clock under this kernel
#endif
#define PHASE(i) do { if (TIMED && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[i] += n_ - tc; tc = n_; tc2 = n_; } } while (0)
#ifdef JAMD_DEV
#define PROBE(g, i) do { if (TIMED && JAMD_XBEAM_PROBE == (g) && tid == 0) { const unsigned long long n_ = wall_clock64(); ph[i] += n_ - tc2; tc2 = n_; } } while (0)
#else
#define PROBE(g, i) ((void)0)
#endif
  int max_tokens = resume ? ss->max_tokens : 1;
  bool stopped = false;
  __syncthreads();

  // The frame's score row goes to LDS by LDS-DMA (no registers, nothing waits on it): the row of frame t + 1 is
  // requested when step C of frame t is done with the buffer, and has landed by the pruning step's first barrier.
  auto row_request = [&](int tt) {
    if (!wk.row_cache || tt >= T) return;
    const float *rg = scores + (size_t)(t_begin + tt - base) * S;
    const int ln = tid & 63;
    for (int b = uni((int)(tid >> 6)) * 64; b < S; b += NT)
      if (b + ln < S) __builtin_amdgcn_global_load_lds((glb_void *)(rg + b + ln), (lds_void *)(rowc + b), 4, 0, 0);
  };
  row_request(resume ? base : (dfa ? 0 : 1));
  int par = 0;      // wide layout: the survivors live at o_sv (0) or at the head of the CUR() area (1), in turns: step E writes the next frame's where this frame's are not
  (void)par;
  for (int t = resume ? base : (dfa ? 0 : 1); t <= (finish ? T : T - 1); t++) {
    tid = tid_now();
#if JAMD_XARGS_RELOAD
    XBEAM_VIEWS(xargs_now());                              // this frame's view of the launch constants (see xargs_now())
#endif
    if constexpr (WIDE) sv.p = reinterpret_cast<u32x4 *>(ub + (par ? wk.o_cur : wk.o_sv));   // (the survivors' two homes: step E)
    const int n_surv = uni(sh.n_surv);
    __syncthreads();
    if (tid == 0) { sh.n_new = 0; sh.n_we = 0; sh.n_arc = 0; sh.we_best = 0ull; sh.maxbits = ord(JAMD_LOG_ZERO); sh.minbits = 0xffffffffu; }
    const bool last = (t == T);
    // ---- 0: dense visiting indices.  Source j owns XW slots for its word-internal transitions and, when it
    //         is a word end that may be followed by a word, one slot per root; sources pruned by score own none.
    //         The same scan numbers the trellis words of the frame in visiting order (save_trellis() is called in
    //         that order, beam.c:2209: the atom array comes out in the reference's creation order).
    int nbits = 0;
    {
      int carry = 0, acarry = sh.n_atom;
      for (int j0 = 0; j0 < n_surv; j0 += NT) {
        const int j = j0 + tid;
        int cnt = 0, isend = 0;
        if (j < n_surv) {
          u32x4 a_, b_;
          sv.quads(j, a_, b_);
          const float sc = __uint_as_float(a_.y); const int sw = (int)b_.z;
          const bool alive = last || (sc > JAMD_LOG_ZERO && !(sc < thr));
          if (alive && !last) cnt = XW + ((sw >= 0 && !wordmode && sw != lx.tail_silwid) ? nroot_x : 0);
          isend = (alive && sw >= 0) ? 1 : 0;
        }
        int ex, ea;
        block_excl_scan2<NT>(sh, cnt, isend, ex, ea);
        if (j < n_surv) { dbase[j] = carry + ex; sv_atom[j] = isend ? acarry + ea : -1; }
        carry += sh.scan_total; acarry += sh.scan_total2;
        __syncthreads();
      }
      if (tid == 0) { dbase[n_surv] = carry; sh.n_atom = acarry; }
      nbits = last ? 0 : carry + (dfa ? (t == 0 ? lx.ninit : 0) : XW + lx.nshared);
    }
    PROBE(1, 4);
    const int nwords = (nbits + 31) >> 5;
    const bool bm_in_lds = nwords <= xw.bm_words;
    // the creation-order bitmap: in LDS, or (a frame with more visiting indices than fit) in the utterance's slice
    lds_u32 *bm_l = (lds_u32 *)(dyn_lds + xw.off_bm);
    unsigned *bm_g = reinterpret_cast<unsigned *>(ub + xw.o_bitmap);
    auto bm_get = [&](int w) -> unsigned { return bm_in_lds ? bm_l[w] : bm_g[w]; };
    __syncthreads();

    // nscid = successor id of next_node (0 for the self loop): the caller loads it beside the transition record
    auto intra_candidate = [&](const Tok &tk, int j, int next_node, float a, int sub, int nscid) {
      float tmpsum = tk.score + a;
      if (nscid != 0) {
        const float ng = max_successor_prob(lx, tk.last_cword, nscid, memo) * lmw + pen;
        tmpsum -= tk.last_lscore;
        tmpsum += ng;
      }
      xpush(sh, cl, next_node, tmpsum, ((unsigned)j << s1) | (unsigned)sub);
    };
    // ---- A: intra-word transitions + word-end atoms (beam.c:2838-2900)
    for (int j = tid; j < n_surv; j += NT) {
      const Tok tk = sv.load(j);
      const int node = tk.node;
      const int sword = tk.pad0;                       // stend
      if (!last) {
        if (tk.score <= JAMD_LOG_ZERO) continue;
        if (tk.score < thr) continue;
        const int4 na = lx.node_a(node);
        const int nscid1 = (node + 1 < lx.nnode) ? lx.scid(node + 1) : 0;   // independent of na: both loads in flight together
        const int e0 = na.z, e1 = na.w;
        if (e1 > e0) {
          const int b0 = atomicAdd(&sh.n_arc, e1 - e0);
          for (int e = e0; e < e1; e++) ARCQ(b0 + e - e0) = make_int2(j | ((2 + e - e0) << 16), e);   // (source | transition number, arc)
        }
        { const float a = __int_as_float(na.x); if (a != JAMD_LOG_ZERO) intra_candidate(tk, j, node, a, 0, 0); }
        { const float a = __int_as_float(na.y); if (a != JAMD_LOG_ZERO) intra_candidate(tk, j, node + 1, a, 1, nscid1); }
      }
      if (sword >= 0) {
        const int ai = sv_atom[j];                         // save_trellis() :2209-2247, numbered in step 0
        if (ai < wk.atom_cap) {
          jamd_trellis_atom a;
          a.wid = sword; a.last_tre = tk.last_tre; a.backscore = tk.score; a.lscore = tk.last_lscore;
          a.begintime = (short)((tk.last_tre < 0 ? -1 : ATOM(tk.last_tre).endtime) + 1);
          a.endtime = (short)(t - 1);
          ATOM(ai) = a;
        }
        if (!last && !wordmode && sword != lx.tail_silwid) {   // beam_inter_word() :2296-2313
          welist[atomicAdd(&sh.n_we, 1)] = j;
          const float tmpprob = tk.score + lx.wordend_a(sword);
          if (!dfa && tmpprob > JAMD_LOG_ZERO)                 // strict < in the reference: the earliest of the best
            atomicMax(&sh.we_best, ((unsigned long long)ordz(tmpprob) << 32) | (unsigned)(~(unsigned)j));
        }
      }
    }
    __syncthreads();
    PROBE(1, 5);
    if (!last) {
      const int n_arc = uni(sh.n_arc);
      for (int q = tid; q < n_arc; q += NT) {
        const int2 it = ARCQ(q);
        const int j = it.x & 0xffff;
        const int to = lx.ac_to(it.y);
        const Tok tk = sv.load(j);
        intra_candidate(tk, j, to, lx.ac_a(it.y), it.x >> 16, to != tk.node ? lx.scid(to) : 0);
      }
      __syncthreads();
    }
    PHASE(0);
    if (last) break;

    // ---- B: cross-word transitions.  Roots are visited from startnum-1 down to 0 (beam.c:2334, :2562); the
    //         root lists of the lexicon image are stored in that order.
    if (dfa) {
      const int n_we = sh.n_we, nroot = lx.startnum;
      const int total = n_we * nroot;
      for (int x = tid; x < total; x += NT) {
        const int w = x / nroot, rv = x - w * nroot;
        const int r = nroot - 1 - rv;
        const int j = welist[w];
        const Tok tk = sv.load(j);
        const int sword = tk.pad0;
        if (!lx.cat_pair(lx.wton(sword) * lx.ncat + lx.root_cat(r))) continue;
        if (lx.nfwd && fwd_next(lx, tk.pad1, lx.root_cat(r)) < 0) continue;     // forward DFA: no arc for this category (:2412-2422)
        const int last_word = lx.is_transparent(sword) ? tk.last_cword : sword;
        float tmpsum = tk.score;
        tmpsum += lx.wordend_a(sword);
        float ng = lx.penalty1;
        ng += (last_word >= 0) ? lx.cprob(last_word) : 0.0f;
        tmpsum += ng;
        xpush(sh, cl, lx.startnode(r), tmpsum, ((unsigned)j << s1) | (unsigned)(XW + rv));
      }
      if (t == 0)
        for (int e = tid; e < lx.ninit; e += NT) xpush(sh, cl, lx.init_node(e), lx.init_lscore(e), (unsigned)e);
    } else {
      // beam_inter_word() :2296-2440, root by root: a word end is reduced to (score + exit transition, LM context,
      // source position) once, then every isolated root takes the best candidate and the earliest visit over the
      // word ends with independent, coalesced reads of the cross-word LM table -- one cell update per root
      // instead of one per (word end, root).  The result is the one of pushing the candidates one by one
      // (a cell keeps a maximum and the first visit a minimum).
      const int n_we = uni(sh.n_we), niso = lx.isolatenum;
      lds_v4 *werec = (lds_v4 *)tpre;                           // [kWeChunk] (tpre is free until step C0)
      constexpr int kWeChunk = NT / 4;
      for (int w0 = 0; w0 < n_we; w0 += kWeChunk) {
        const int nrec = min(kWeChunk, n_we - w0);
        if (tid < nrec) {
          const int j = welist[w0 + tid];
          const Tok tk = sv.load(j);
          const int sword = tk.pad0;
          const bool tr = lx.is_transparent(sword) != 0;
          const int last_word = tr ? tk.last_cword : sword;
          float bs = tk.score;
          bs += lx.wordend_a(sword);
          const bool trans2 = tr && tk.last_cword >= 0 && lx.is_transparent(tk.last_cword);
          u32x4 rec;
          rec.x = __float_as_uint(bs); rec.y = (unsigned)(last_word < 0 ? -1 : lx.wton(last_word));
          rec.z = (unsigned)j | (trans2 ? 0x80000000u : 0u); rec.w = (unsigned)last_word;
          werec[tid] = rec;
        }
        __syncthreads();
        int parts = NT / (niso > 0 ? niso : 1);
        if (parts < 1) parts = 1;
        if (parts > nrec) parts = nrec;
        const int total = niso * parts;
        for (int x = tid; x < total; x += NT) {
          const int part = x / niso, i = x - part * niso;
          const int2 ir = lx.iso_root(i);
          unsigned long long best = 0ull; unsigned nfirst = 0u;
          // four word ends at a time: the table reads of a group go out together (one memory latency per group, not
          // per word end; a maximum and a minimum do not care about the order)
          constexpr int G = 4;
          for (int w0g = part; w0g < nrec; w0g += G * parts) {
            u32x4 rec[G]; float p[G]; bool live[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
              const int w = w0g + g * parts;
              live[g] = w < nrec;
              rec[g] = werec[live[g] ? w : part];
              const int ctx = (int)rec[g].y;
              p[g] = (!live[g] || ctx < 0) ? 0.0f
                     : lx.iwtab ? lx.iwtab[(size_t)ctx * niso + i]
                     : bigram_prob(lx, ctx, lx.wton(ir.y)) + lx.cprob(ir.y);
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
              if (!live[g]) continue;
              float tmpsum = __uint_as_float(rec[g].x);
              const float ng = p[g] * lmw + pen;
              tmpsum += ng;
              if (rec[g].z & 0x80000000u) tmpsum += lx.lm_penalty_trans;
              if (tmpsum <= JAMD_LOG_ZERO) continue;
              const unsigned nv = ~(((rec[g].z & 0x7fffffffu) << s1) | (unsigned)(XW + i));
              const unsigned long long key = ((unsigned long long)ordz(tmpsum) << 32) | nv;
              if (key > best) best = key;
              if (nv > nfirst) nfirst = nv;
            }
          }
          if (best != 0ull) xpush_key(sh, cl, ir.x, best, nfirst);
        }
        __syncthreads();
      }
      PROBE(1, 7);
      if (sh.we_best != 0ull) {                       // beam_inter_word_factoring() :2549-2637
        const unsigned long long kb = sh.we_best;
        const float best_score = unord((unsigned)(kb >> 32));
        const Tok tk = sv.load((int)(~(unsigned)kb));
        const int sword = tk.pad0;
        const bool trans2 = lx.is_transparent(sword) && tk.last_cword >= 0 && lx.is_transparent(tk.last_cword);
        for (int r = tid; r < lx.nshared; r += NT) {
          const float2 sr = lx.shared_root(r);
          const float ng = sr.y * lmw + pen;
          float tmpsum = best_score;
          tmpsum += ng;
          if (trans2) tmpsum += lx.lm_penalty_trans;
          if (tmpsum < thr) continue;                               // :2580
          xpush(sh, cl, __float_as_int(sr.x), tmpsum, ((unsigned)n_surv << s1) | (unsigned)(XW + r));
        }
      }
    }
    __syncthreads();
    if (tid == 0) sh.n_arc = 0;
    PHASE(1);

    // ---- C0: creation order = rank of the node's first visit (create_token() :1148)
    const int n_new = uni(sh.n_new);
    if (n_new > max_tokens) max_tokens = n_new;
    if (pm.pstat && tid == 0) { pm.pstat[8] += n_new; pm.pstat[9] += n_surv; pm.pstat[10] += sh.n_we; pm.pstat[11] += 1; }   // work counters (jamd_beam_prune_stats())
    if (n_new > wk.tok_cap) {              // cannot happen (tok_cap bounds the reachable nodes); never write past the arrays
      if (tid == 0) res->status = JAMD_PASS1_OVERFLOW;
      stopped = true;
      __syncthreads();
      break;
    }
    int Wsh = 0;                                      // bitmap words per thread in the prefix scan: a power of two
    while ((NT << Wsh) < nwords) Wsh++;
    const int W = 1 << Wsh;
    {
      for (int i = tid; i < nwords; i += NT) { if (bm_in_lds) bm_l[i] = 0u; else bm_g[i] = 0u; }
      __syncthreads();
      for (int s = tid; s < n_new; s += NT) {
        const int2 t2 = TOUCHED(s);
        const unsigned fv = ~(t2.y >= 0 ? cl.lfirst[t2.y] : NODEFIRST(t2.x));
        const int dense = ((dfa && t == 0) ? 0 : dbase[fv >> s1]) + (int)(fv & submask);
        if (bm_in_lds) atomicOr((unsigned *)&bm_l[dense >> 5], 1u << (dense & 31)); else atomicOr(&bm_g[dense >> 5], 1u << (dense & 31));
      }
      __syncthreads();
      PROBE(2, 4);
      int cnt = 0;
      for (int x = 0; x < W; x++) { const int w = tid * W + x; if (w < nwords) cnt += __popc(bm_get(w)); }
      const int ex = block_excl_scan<NT>(sh, cnt);
      tpre[tid] = (unsigned)ex;
      __syncthreads();
      PROBE(2, 5);
    }
    // ---- C: finalize the touched nodes: winner's payload + acoustic score (:2944-2951), stored at the
    //         token's creation index
    {
      const XRowRef row{scores + (size_t)(t_begin + t - base) * S, rowc, wk.row_cache != 0};
      unsigned mymax = ord(JAMD_LOG_ZERO), mymin = 0xffffffffu;
      constexpr int CB = JAMD_XBEAM_CB;
      for (int s0 = tid; s0 < n_new; s0 += CB * NT) {
        bool ok[CB]; int node[CB], slot[CB], tokid[CB]; int4 nr[CB]; unsigned long long key[CB]; unsigned fvis[CB];
        int l_tre[CB], l_wid[CB], ent[CB];
#pragma unroll
        for (int k = 0; k < CB; k++) {
          const int s = s0 + k * NT;
          ok[k] = s < n_new;
          const int2 t2 = ok[k] ? TOUCHED(s) : make_int2(0, -1);
          node[k] = t2.x; slot[k] = t2.y;
        }
#pragma unroll
        for (int k = 0; k < CB; k++) nr[k] = lx.node_b(node[k]);
#pragma unroll
        for (int k = 0; k < CB; k++) {
          key[k] = 0ull; fvis[k] = 0u;
          if (ok[k]) {
            if (slot[k] >= 0) {
              key[k] = cl.lkey[slot[k]]; fvis[k] = ~cl.lfirst[slot[k]];
              cl.lkey[slot[k]] = 0ull; cl.lnode[slot[k]] = -1; cl.lfirst[slot[k]] = 0u;
            } else {
              key[k] = atomicExch(&NODEKEY(node[k]), 0ull);
              fvis[k] = ~atomicExch(&NODEFIRST(node[k]), 0u);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < CB; k++) {
          tokid[k] = 0;
          if (!ok[k]) continue;
          const int dense = ((dfa && t == 0) ? 0 : dbase[fvis[k] >> s1]) + (int)(fvis[k] & submask);
          const int w = dense >> 5, tw = w >> Wsh;
          int r = (int)tpre[tw];
          for (int x = tw * W; x < w; x++) r += __popc(bm_get(x));
          r += __popc(bm_get(w) & ((1u << (dense & 31)) - 1u));
          tokid[k] = r;
        }
        // of the winner's payload only what the score needs (the word whose last phone selects the state of a word-head
        // node) and what the pruning step overwrites (the trellis word a cross-word winner comes from); the rest is built
        // for the SURVIVORS after the pruning step, from the record {node, visiting index, trellis word, score} (round 6)
#pragma unroll
        for (int k = 0; k < CB; k++) {
          const unsigned vis = ~(unsigned)key[k];
          l_tre[k] = -2; l_wid[k] = -1;
          if (!ok[k]) continue;
          int j = (int)(vis >> s1);
          const int sub = (int)(vis & submask);
          if (dfa && t == 0) { l_tre[k] = -1; continue; }
          if (j < n_surv && sub < XW) {                          // intra-word: inherited
            if (nr[k].w >= JAMD_AS_RSET) l_wid[k] = sv.load(j).last_wid;
          } else {
            if (!(j < n_surv)) j = (int)(~(unsigned)sh.we_best);  // the factoring pass: from the best word end
            l_tre[k] = sv_atom[j];
            if (nr[k].w >= JAMD_AS_RSET) l_wid[k] = sv.load(j).pad0;
          }
        }
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
          const int s = tokid[k];
          const float score = unord((unsigned)(key[k] >> 32));
          float sc = score;
          if (ent[k] >= 0) {
            sc = score + ac[k];
            const unsigned b = ordz(sc);
            CURKEY(s) = b;
            if (b > mymax) mymax = b;
            if (b < mymin) mymin = b;
          } else {
            ARCQ(atomicAdd(&sh.n_arc, 1)) = make_int2(s, ~ent[k]);
          }
          REC(s) = u32x4{(unsigned)node[k], ~(unsigned)key[k], (unsigned)l_tre[k], __float_as_uint(sc)};
        }
      }
      __syncthreads();
      PROBE(2, 6);
      // state-set reductions (outprob_cd(), outprob.c:287-400): four, two or one lane per (token, set), so that the
      // frame's sets go through in one round when they can; eight member loads in flight per lane
      const int n_set = uni(sh.n_arc);
      const int lps = n_set <= NT / 4 ? 4 : (n_set <= NT / 2 ? 2 : 1), lsh = lps == 4 ? 2 : (lps == 2 ? 1 : 0);
      const int sub = tid & (lps - 1), lane = tid & 63;
      for (int q0 = 0; q0 < n_set; q0 += NT >> lsh) {
        const int q = q0 + (tid >> lsh);
        const bool act = q < n_set;
        const int2 it = act ? ARCQ(q) : make_int2(0, 0);
        const int a = act ? lx.set_off(it.y) : 0, bnd = act ? lx.set_off(it.y + 1) : 0;
        const float sc0 = (act && sub == 0) ? __uint_as_float(REC(it.x).w) : 0.0f;      // in flight beside the member loads
        float r;
        if (lx.cdset_method == JAMD_IWCD_NBEST && lx.cdmax_num <= 4) {
          r = lx.cdmax_num <= 3 ? nbest_of_set<3>(lx, row, a, bnd, sub, lps) : nbest_of_set<4>(lx, row, a, bnd, sub, lps);
        } else if (lx.cdset_method == JAMD_IWCD_MAX) {
          float m_ = JAMD_LOG_ZERO;
          for (int m = a + sub; m < bnd; m += 8 * lps) {
            int ix[8]; float pv[8];
#pragma unroll
            for (int jj = 0; jj < 8; jj++) ix[jj] = (m + lps * jj < bnd) ? lx.set_states(m + lps * jj) : -1;
#pragma unroll
            for (int jj = 0; jj < 8; jj++) pv[jj] = (ix[jj] >= 0) ? row[ix[jj]] : JAMD_LOG_ZERO;
#pragma unroll
            for (int jj = 0; jj < 8; jj++) if (m_ < pv[jj]) m_ = pv[jj];
          }
          for (int src = 1; src < lps; src++) { const float c = __shfl(m_, (lane & ~(lps - 1)) + src, 64); if (m_ < c) m_ = c; }
          r = m_;
        } else {
          r = (act && sub == 0) ? cd_reduce(row, lx.set_states_ptr(), a, bnd, lx.cdset_method, lx.cdmax_num) : 0.0f;
        }
        if (act && sub == 0) {
          const float sc = sc0 + r;
          REC(it.x).w = __float_as_uint(sc);
          const unsigned b = ordz(sc);
          CURKEY(it.x) = b;
          if (b > mymax) mymax = b;
          if (b < mymin) mymin = b;
        }
      }
      atomicMax(&sh.maxbits, mymax);
      atomicMin(&sh.minbits, mymin);
    }
    __syncthreads();
    row_request(t + 1);
    PHASE(2);
    {
      const float mx = unord(sh.maxbits);
      thr = (wk.width >= 0.0f) ? (mx - wk.width) : JAMD_LOG_ZERO;
      if (t == 0) thr = JAMD_LOG_ZERO;
    }
    if (n_new == 0) {
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
    // ---- D: rank pruning with the reference's heap; the next frame visits sv[0..n_keep) in this order
    const int n_keep = exact_prune<WIDE, NT>(sh, &CURKEY(0), n_new, wk.beam, Hlds, xw.heap_cap, Hglob, pm, welist, xw.prune_mode, Gcol, (TIMED && (JAMD_XBEAM_PROBE == 0 || JAMD_XBEAM_PROBE == 3 || JAMD_XBEAM_PROBE == 5 || JAMD_XBEAM_PROBE == 6)) ? ph : nullptr);
    // ---- E: the survivors' records (create_token() / propagate_token(): TOKEN2's last_tre, last_cword, last_lscore ...).
    //         Only the <= beam tokens that are kept get one: step C left {node, visiting index, trellis word, score} for every
    //         token (16 bytes instead of 32), and the sources' records and the LM look-ups are read for the survivors only.
    {
      constexpr int CB = JAMD_XBEAM_CB;
      for (int s0 = tid; s0 < n_keep; s0 += CB * NT) {
        bool ok[CB]; int node[CB]; int4 nr[CB]; u32x4 rec[CB];
        int l_tre[CB], l_cword[CB], l_wid[CB], lmreq[CB], l_to[CB];       // l_to: forward-DFA state (TOKEN2.to_state), 0 without one
        float l_ls[CB];
#pragma unroll
        for (int k = 0; k < CB; k++) {
          const int jn = s0 + k * NT;
          ok[k] = jn < n_keep;
          rec[k] = ok[k] ? REC(welist[jn]) : u32x4{0u, 0u, 0u, 0u};
          node[k] = (int)rec[k].x;
        }
#pragma unroll
        for (int k = 0; k < CB; k++) nr[k] = lx.node_b(node[k]);
        // the winner's payload from its visiting index (the sources are the OLD survivors: the new ones go through CUR())
#pragma unroll
        for (int k = 0; k < CB; k++) {
          const unsigned vis = rec[k].y;
          lmreq[k] = 0; l_tre[k] = -1; l_cword[k] = -1; l_wid[k] = -1; l_ls[k] = 0.0f; l_to[k] = 0;
          if (!ok[k]) continue;
          int j = (int)(vis >> s1);
          const int sub = (int)(vis & submask);
          if (dfa && t == 0) {                                 // an initial token of the grammar
            l_ls[k] = lx.init_lscore(sub);
            if (lx.nfwd) l_to[k] = lx.init_to_state(sub);      // :1739-1747
          } else if (j < n_surv && sub < XW) {                 // intra-word
            const Tok tk = sv.load(j);
            l_tre[k] = tk.last_tre; l_cword[k] = tk.last_cword; l_wid[k] = tk.last_wid; l_to[k] = tk.pad1;   // (:2120: the state is inherited)
            if (node[k] != tk.node && nr[k].y != 0) lmreq[k] = nr[k].y;   // beam_intra_word_core() :2069-2082
            else l_ls[k] = tk.last_lscore;
          } else {
            const bool iso = j < n_surv;
            if (!iso) j = (int)(~(unsigned)sh.we_best);        // the factoring pass: from the best word end
            const Tok tk = sv.load(j);
            const int sword = tk.pad0;
            const int last_word = lx.is_transparent(sword) ? tk.last_cword : sword;
            l_tre[k] = (int)rec[k].z; l_cword[k] = last_word; l_wid[k] = sword;
            if (dfa) {                                       // beam_inter_word() :2452-2461
              float ng = lx.penalty1;
              ng += (last_word >= 0) ? lx.cprob(last_word) : 0.0f;
              l_ls[k] = ng;
              if (lx.nfwd) l_to[k] = fwd_next(lx, tk.pad1, lx.root_cat(lx.startnum - 1 - (sub - XW)));   // the arc step B found (:2415-2420)
            } else if (iso) {                                // beam_inter_word() :2430-2438
              float p = 0.0f;
              if (last_word >= 0) {
                if (lx.iwtab) p = lx.iwtab[(size_t)lx.wton(last_word) * lx.isolatenum + (sub - XW)];
                else { const int wn = lx.scword(nr[k].y); p = bigram_prob(lx, lx.wton(last_word), lx.wton(wn)) + lx.cprob(wn); }
              }
              l_ls[k] = p * lmw + pen;
            } else {                                         // beam_inter_word_factoring() :2572-2573
              l_ls[k] = lx.fscore(-nr[k].y) * lmw + pen;
            }
          }
        }
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
            float p = 0.0f;
            if (l_cword[k] >= 0) {
              if (lmreq[k] < 0) p = fs[k];
              else if ((int)(unsigned)(mm[k] >> 32) == ctx[k]) p = __uint_as_float((unsigned)mm[k]);
              else p = max_successor_prob(lx, l_cword[k], lmreq[k], memo);
            }
            l_ls[k] = p * lmw + pen;
          }
        }
#pragma unroll
        for (int k = 0; k < CB; k++) {
          if (!ok[k]) continue;
          Tok nw;
          nw.node = node[k]; nw.score = __uint_as_float(rec[k].w); nw.pad0 = nr[k].x; nw.pad1 = l_to[k];
          nw.last_tre = l_tre[k]; nw.last_cword = l_cword[k]; nw.last_wid = l_wid[k]; nw.last_lscore = l_ls[k];
          if constexpr (WIDE) { XSv<true> svn; svn.p = reinterpret_cast<u32x4 *>(ub + (par ? wk.o_sv : wk.o_cur)); svn.store(s0 + k * NT, nw); }
          else CUR(s0 + k * NT) = nw;
        }
      }
    }
    __syncthreads();
    if constexpr (WIDE) par ^= 1;
    else { for (int j = tid; j < n_keep; j += NT) sv.store(j, CUR(j)); }   // (LDS: the sources had to be read first)
    if (tid == 0) sh.n_surv = n_keep;
    // the pruning step used the cell area: empty it again
    for (int i = tid; i < cl.nslot; i += NT) { cl.lkey[i] = 0ull; cl.lnode[i] = -1; cl.lfirst[i] = 0u; }
    __syncthreads();
    PHASE(3);
  }
  __syncthreads();

  if (smode == 1) {
    if constexpr (!WIDE) {
      if (!stopped) {
        u32x4 *dst = (u32x4 *)(ub + wk.o_sv);
        for (int i = tid; i < wk.sv_bytes / 16; i += NT) dst[i] = sv.p[i];
      }
    } else {
      if (!stopped && par) {                                 // the next launch finds the survivors at o_sv
        const u32x4 *src = (const u32x4 *)(ub + wk.o_cur);
        u32x4 *dst = (u32x4 *)(ub + wk.o_sv);
        for (int i = tid; i < 2 * sh.n_surv; i += NT) dst[i] = src[i];
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

  // ---- find_1pass_result() :399-431 + trace_backptr() :294-340
  const int natom = min(sh.n_atom, wk.atom_cap);
  if (tid == 0) sh.best_atom = -1;
  __syncthreads();
  if (res->status == JAMD_PASS1_OK && dfa) {
    // grammar / word list (:433-455): the best word on the latest frame that has one.  The reference walks rw[t], which
    // bt_sort_rw() has sorted by word id, with a strict <: of equally good words the smaller id wins -- the key below.
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
    // the tail-silence word ending latest; atoms of one frame are emitted together, so "latest" is
    // decided on the end time, not on the index
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
#ifdef JAMD_DEV
    if (TIMED && JAMD_XBEAM_PROBE == 4) res->phase_us[7] = (int)((clock64() - cyc0) * 100ull / (wall_clock64() - wall0));   // MHz
#endif
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

#include "beam_exact_mp.h"

// diagnostic: the pruning step alone on given score bits (tests/test_prune_order.py fuzzes it against the
// sequential heap)
template <bool WIDE, int NT, bool FULL>
__global__ void __launch_bounds__(NT) prune_order_kernel(XWork xw, const unsigned *keys, int n, int k, int *out, int *nout,
                                                         unsigned long long *hglob, u32x4 *gcol, unsigned char *gsweep, int *arr) {
  __shared__ XShared sh;
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  PruneMem pm;
  pm.compR = (lds_u64 *)(dyn_lds + xw.off_compr); pm.compT = pm.compR + xw.b_cap;
  pm.vposR = (lds_u32 *)(dyn_lds + xw.off_vpos);
  pm.idR = (lds_u32 *)(dyn_lds + xw.off_id);
  pm.idT = (lds_u32 *)(dyn_lds + xw.off_idt);
  pm.hist = (lds_u32 *)(dyn_lds + xw.off_hist);
  pm.tailmask = (lds_u32 *)(dyn_lds + xw.off_tail);
  pm.cand = (lds_i32 *)(pm.tailmask + (xw.w.beam + 31) / 32 + 2);
  pm.occ = pm.cand + kMaxCand; pm.need = pm.occ + kMaxCand; pm.takers = pm.need + kMaxCand + 4;
  pm.ordv = pm.takers + (kMaxCand + 1) * (kTakers + 1);
  pm.b_cap = xw.b_cap;
  pm.sw_region = (unsigned char JAMD_LDS *)(dyn_lds + xw.off_dov); pm.sw_bytes = xw.off_row - xw.off_dov;
  pm.sw_glob = gsweep;
  pm.pstat = nullptr;
  lds_i32 *svid = (lds_i32 *)(dyn_lds + xw.off_we);
  unsigned mx = 0u, mn = 0xffffffffu;
  for (int i = threadIdx.x; i < n; i += NT) { const unsigned b = keys[i]; if (b > mx) mx = b; if (b < mn) mn = b; }
  if (threadIdx.x == 0) { sh.maxbits = 0u; sh.minbits = 0xffffffffu; sh.sw_info = 0; sh.sw_ticks = 0; sh.sw_nev_out = 0; for (int i = 0; i < 8; i++) sh.sw_prof[i] = 0; for (int i = 0; i < 4; i++) sh.df_prof[i] = 0; }
  __syncthreads();
  atomicMax(&sh.maxbits, mx); atomicMin(&sh.minbits, mn);
  __syncthreads();
  const int nk = exact_prune<WIDE, NT, FULL>(sh, keys, n, k, (lds_u64 *)(dyn_lds + xw.off_heap), xw.heap_cap, hglob, pm, svid,
                                         xw.prune_mode, gcol, nullptr, arr);
  for (int j = threadIdx.x; j < nk; j += NT) out[j] = svid[j];
  if (threadIdx.x == 0) { nout[0] = nk; nout[1] = sh.sw_info; nout[2] = sh.sw_ticks; nout[3] = sh.sw_nev_out; for (int i = 0; i < 8; i++) nout[4 + i] = sh.sw_prof[i]; for (int i = 0; i < 4; i++) nout[12 + i] = sh.df_prof[i]; }
}

}  // namespace

namespace jamdb {

// The fixed part of the image for one of the two layouts.
//   narrow: [survivors Tok[beam]] [atom] [welist] [dbase] [tpre] [bitmap] | cells / pruning overlay | score row
//   wide:   [welist] | [atom] [dbase] [tpre] [bitmap] cells | score row     -- the survivors live in the utterance's
//           slice (o_sv: steps 0 and A read them in order, only the winner look-ups of step C are gathers), and the
//           pruning step overlays everything behind welist[] (all of it is dead between step C and the next step 0;
//           welist[] carries the pruning step's result).
static int xbeam_fixed(XWork *xw, bool wide, int maxfan, int nroot, int ninit, int nshared) {
  const int beam = xw->w.beam;
  int at = 0;
  auto place = [&](int *off, int bytes) { *off = at; at = (at + bytes + 15) & ~15; };
  xw->wide = wide ? 1 : 0;
  if (!wide) {
    at = beam * (int)sizeof(Tok);
    place(&xw->off_atom, 4 * beam);
    place(&xw->off_we, 4 * beam);
    place(&xw->off_dbase, 4 * (beam + 2));
    xw->w.sv_bytes = at;                             // what a streaming session parks between launches
    place(&xw->off_tpre, 4 * xw->nt);
  } else {
    place(&xw->off_we, 4 * beam);
    xw->off_dov = at;
    place(&xw->off_atom, 4 * beam);
    place(&xw->off_dbase, 4 * (beam + 2));
    xw->w.sv_bytes = (beam * (int)sizeof(Tok) + 15) & ~15;   // the survivors' home in the slice; nothing to park
    place(&xw->off_tpre, 4 * xw->nt);
  }
  if (at + 8 * 1024 > xw->lds_budget) return -2;
  // creation-order bitmap: XW bits per source plus a few word ends' worth of roots (a frame that needs more
  // uses the copy in global memory); at most an eighth of what is left
  int bm_words = (beam * maxfan + 8 * nroot + nshared + ninit + 31) / 32 + 64;
  if (bm_words > 4096) bm_words = 4096;
  if (4 * bm_words > (xw->lds_budget - at) / 8) bm_words = (xw->lds_budget - at) / 32;
  xw->bm_words = bm_words;
  place(&xw->off_bm, 4 * bm_words);
  xw->cells_at = at;
  if (!wide) xw->off_dov = at;
  return 0;
}

static int xbeam_tail_bytes(int beam) {
  return (4 * ((beam + 31) / 32 + 2 + 4 * kMaxCand + 4 + (kMaxCand + 1) * (kTakers + 1)) + 15) & ~15;
}

// The per-launch part with `want` bytes set aside for the score row.  The frame's Viterbi cells take what is left
// (16 bytes a slot); the pruning step overlays them (narrow) or everything behind welist[] (wide):
//   narrow: [compR|compT 16 b_cap] [vposR 4] [idR 4] [hist] [tail] [heap: the rest]
//   wide:   [compA|compB 16 b_cap] [idA 4] [idB 4] ... [hist] [tail]   with the heap laid over the lists (it is dead
//           once the top elements are collected into o_collect) and vposR over compA (dead once the list is sorted)
// b_cap = 0: no room for the closed-form extraction (the sequential extraction runs on one lane).
static void xbeam_place_with(XWork *xw, int want) {
  const int beam = xw->w.beam;
  const int cells_at = xw->cells_at;
  int region = ((xw->lds_budget - cells_at) & ~1023) - want;
  if (region < 0) region = 0;
  int nslot = (region / 16) & ~63;
  if (nslot < 1024) nslot = 0;                       // too few to be worth probing: every cell in nodekey[]
  xw->nslot = nslot;
  xw->off_cells = cells_at;
  xw->off_lnode = cells_at + 8 * nslot;
  xw->off_lfirst = cells_at + 12 * nslot;
  const int end = cells_at + region;
  const int tail_bytes = xbeam_tail_bytes(beam);
  int at = xw->off_dov;
  auto place = [&](int *off, int bytes) { *off = at; at = (at + bytes + 15) & ~15; };
  xw->b_cap = beam + 256;
  if (!xw->wide) {
    if (16 * xw->b_cap + 8 * xw->b_cap + 4 * 2048 + tail_bytes + 128 + 8 * (2 * beam + 64) > region) xw->b_cap = 0;
    place(&xw->off_compr, 16 * xw->b_cap);
    place(&xw->off_vpos, 4 * xw->b_cap);
    place(&xw->off_id, 4 * xw->b_cap);
    xw->off_idt = xw->off_id;
    place(&xw->off_hist, xw->b_cap ? 4 * 2048 : 0);
    place(&xw->off_tail, tail_bytes);
    place(&xw->off_heap, 0);
    xw->heap_cap = (end - xw->off_heap) / 8 - 2;
  } else {
    const int dreg = end - xw->off_dov;
    if (24 * xw->b_cap + 4 * 2048 + tail_bytes + 64 > dreg) xw->b_cap = 0;
    xw->off_tail = end - tail_bytes;
    xw->off_hist = xw->off_tail - 4 * 2048;
    xw->off_heap = xw->off_dov;
    place(&xw->off_compr, 16 * xw->b_cap);
    place(&xw->off_idt, 4 * xw->b_cap);
    place(&xw->off_id, 4 * xw->b_cap);
    xw->off_vpos = xw->off_compr;
    xw->heap_cap = (xw->off_hist - xw->off_heap) / 8 - 2;
  }
  if (xw->heap_cap < 0) xw->heap_cap = 0;
  xw->off_row = end;
  xw->lds_bytes = end;
}

void xbeam_place(XWork *xw, int nstate) {
  xbeam_place_with(xw, 0);
  xw->w.row_cache = 0;
  if (nstate <= 0) return;
  // make room for the frame's score row when the cell table, the LDS heap and the top lists can spare it (the half
  // shape asks for the narrow layout's cell count: with half the LDS, cells lost to the row cost more than the row saves)
  XWork t = *xw;
  xbeam_place_with(&t, (4 * nstate + 1023) & ~1023);
  const int beam = xw->w.beam;
  const bool ok = t.b_cap == xw->b_cap && t.off_row + 4 * nstate <= xw->lds_budget &&
                  (xw->wide && xw->nt == NT ? 2 * t.heap_cap >= 5 * beam : (t.nslot >= 6 * beam && t.heap_cap >= 5 * beam));
  if (!ok) return;                                   // the row stays in global memory
  *xw = t;
  xw->w.row_cache = 1;
}

int xbeam_layout(XWork *xw, const Work &w, int maxfan, int nroot, int ninit, int nshared, bool half, bool mp) {
  xw->w = w;
  xw->mp = mp ? 1 : 0;
  xw->nt = half ? kHalfNT : NT;
  xw->lds_budget = half ? kHalfDynLds : kMaxDynLds;
#ifdef JAMD_DEV
  if (half) { const char *kb = getenv("JAMD_HALF_LDS_KB"); if (kb && atoi(kb) >= 32 && atoi(kb) * 1024 <= kHalfDynLds) xw->lds_budget = atoi(kb) * 1024; }
#endif
  const int beam = w.beam;
  xw->xw = maxfan;                                   // self, next, extra arcs
  int need = maxfan + nroot;                         // transition numbers of one source
  if (ninit > need) need = ninit;
  if (maxfan + nshared > need) need = maxfan + nshared;
  if (mp) {                                          // second half of a multipath frame: root number * maxfan + the root's transition
    if (nroot * maxfan > need) need = nroot * maxfan;
    if (nshared * maxfan > need) need = nshared * maxfan;
  }
  int s1 = 1; while ((1 << s1) < need + 1) s1++;
  int jb = 1; while ((1 << jb) < beam + 2) jb++;
  if (s1 + jb > 32) return -1;
  xw->s1 = s1;
  if ((long long)w.tok_cap + 2 >= (1ll << (kMaxL + 1))) return -3;   // prekey() numbers heap positions below 2^(kMaxL+1)
  // the narrow layout (survivors in LDS) while it leaves room for the closed-form extraction and for the heap of a
  // typical frame (three to six tokens per survivor) beside the top lists, else the wide one
  int rc = half ? -2 : xbeam_fixed(xw, false, maxfan, nroot, ninit, nshared);   // (the half shape: always the wide layout)
  if (rc == 0) { xbeam_place_with(xw, 0); if (xw->b_cap == 0 || xw->heap_cap < 8 * beam) rc = -2; }
  if (rc != 0) {
    rc = xbeam_fixed(xw, true, maxfan, nroot, ninit, nshared);
    if (rc != 0) return rc;
    xbeam_place_with(xw, 0);
  }
  // the half shape is there for throughput: only where a typical frame still runs out of LDS
  if (half && (xw->b_cap == 0 || xw->heap_cap < 5 * beam || xw->nslot < 3 * beam)) return -2;
  xw->w.row_cache = 0;
  xw->prune_mode = 0;
  return 0;
}

hipError_t xbeam_prepare() {
  const void *fn[] = {(const void *)beam_exact_kernel<false, false, NT>, (const void *)beam_exact_kernel<true, false, NT>,
                      (const void *)beam_exact_kernel<false, true, NT>, (const void *)beam_exact_kernel<true, true, NT>,
                      (const void *)beam_exact_mp_kernel<false, false, NT>, (const void *)beam_exact_mp_kernel<true, false, NT>,
                      (const void *)beam_exact_mp_kernel<false, true, NT>, (const void *)beam_exact_mp_kernel<true, true, NT>,
                      (const void *)prune_order_kernel<false, NT, false>, (const void *)prune_order_kernel<true, NT, false>,
                      (const void *)prune_order_kernel<false, NT, true>, (const void *)prune_order_kernel<true, NT, true>};
  for (const void *f : fn) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    if (e != hipSuccess) return e;
  }
  // the half shape is always the wide layout (xbeam_layout())
  const void *fh[] = {(const void *)beam_exact_kernel<false, true, kHalfNT>, (const void *)beam_exact_kernel<true, true, kHalfNT>,
                      (const void *)beam_exact_mp_kernel<false, true, kHalfNT>, (const void *)beam_exact_mp_kernel<true, true, kHalfNT>,
                      (const void *)prune_order_kernel<true, kHalfNT, false>};
  for (const void *f : fh) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kHalfDynLds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

void xbeam_launch(const LexDev &lx, const XWork &xw0, const float *scores, int nstate, const int *d_utt_off, int nutt,
                  int smode, bool timed, hipStream_t st) {
  XWork xw = xw0;
  xbeam_place(&xw, nstate);
  const int lds = xw.lds_bytes + (xw.w.row_cache ? 4 * nstate : 0);
  const dim3 grid(nutt), block(xw.nt);
#define JAMD_XLAUNCH(W, N)                                                                                                   \
  do {                                                                                                                       \
    if (timed) hipLaunchKernelGGL((beam_exact_kernel<true, W, N>), grid, block, lds, st, XKArgs{lx, xw}, scores, nstate, d_utt_off, smode); \
    else hipLaunchKernelGGL((beam_exact_kernel<false, W, N>), grid, block, lds, st, XKArgs{lx, xw}, scores, nstate, d_utt_off, smode);      \
  } while (0)
  if (xw.mp) {                                         // multipath lexicons: their own frame (beam_exact_mp.h)
    if (xw.nt == kHalfNT) {                              // half shape (round 5): always the wide layout
      if (timed) hipLaunchKernelGGL((beam_exact_mp_kernel<true, true, kHalfNT>), grid, block, lds, st, XKArgs{lx, xw}, scores, nstate, d_utt_off, smode);
      else hipLaunchKernelGGL((beam_exact_mp_kernel<false, true, kHalfNT>), grid, block, lds, st, XKArgs{lx, xw}, scores, nstate, d_utt_off, smode);
    } else if (xw.wide) {
      if (timed) hipLaunchKernelGGL((beam_exact_mp_kernel<true, true, NT>), grid, block, lds, st, XKArgs{lx, xw}, scores, nstate, d_utt_off, smode);
      else hipLaunchKernelGGL((beam_exact_mp_kernel<false, true, NT>), grid, block, lds, st, XKArgs{lx, xw}, scores, nstate, d_utt_off, smode);
    } else {
      if (timed) hipLaunchKernelGGL((beam_exact_mp_kernel<true, false, NT>), grid, block, lds, st, XKArgs{lx, xw}, scores, nstate, d_utt_off, smode);
      else hipLaunchKernelGGL((beam_exact_mp_kernel<false, false, NT>), grid, block, lds, st, XKArgs{lx, xw}, scores, nstate, d_utt_off, smode);
    }
  }
  else if (xw.nt == kHalfNT) JAMD_XLAUNCH(true, kHalfNT);
  else if (xw.wide) JAMD_XLAUNCH(true, NT);
  else JAMD_XLAUNCH(false, NT);
#undef JAMD_XLAUNCH
}

void xbeam_prune_order_launch(const XWork &xw, const unsigned *d_keys, int n, int k, int *d_out, int *d_nout,
                              unsigned long long *d_hglob, u32x4 *d_collect, unsigned char *d_sweep, int *d_arr, hipStream_t st) {
  if (d_arr) {                                          // the whole array (exact_prune<FULL>: full shape only)
    if (xw.wide) hipLaunchKernelGGL((prune_order_kernel<true, NT, true>), dim3(1), dim3(NT), xw.lds_bytes, st, xw, d_keys, n, k, d_out, d_nout, d_hglob, d_collect, d_sweep, d_arr);
    else hipLaunchKernelGGL((prune_order_kernel<false, NT, true>), dim3(1), dim3(NT), xw.lds_bytes, st, xw, d_keys, n, k, d_out, d_nout, d_hglob, d_collect, d_sweep, d_arr);
  }
  else if (xw.nt == kHalfNT) hipLaunchKernelGGL((prune_order_kernel<true, kHalfNT, false>), dim3(1), dim3(kHalfNT), xw.lds_bytes, st, xw, d_keys, n, k, d_out, d_nout, d_hglob, d_collect, d_sweep, nullptr);
  else if (xw.wide) hipLaunchKernelGGL((prune_order_kernel<true, NT, false>), dim3(1), dim3(NT), xw.lds_bytes, st, xw, d_keys, n, k, d_out, d_nout, d_hglob, d_collect, d_sweep, nullptr);
  else hipLaunchKernelGGL((prune_order_kernel<false, NT, false>), dim3(1), dim3(NT), xw.lds_bytes, st, xw, d_keys, n, k, d_out, d_nout, d_hglob, d_collect, d_sweep, nullptr);
}

size_t xbeam_sweep_bytes(int beam) { return sweep_global_bytes(beam + 256); }

}  // namespace jamdb
