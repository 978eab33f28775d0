// beam_sweep.h -- the events of the extraction loop of sort_token_upward() (libjulius/src/beam.c:1368-1383), ALL AT ONCE.
// Included by beam_exact.hip behind its helpers (block scans, prekey(), tid_now()).
//
// beam_exact.hip replaces the extraction loop by its closed form: the top elements come out sorted by (score, pre-order
// of their heap position), except for "events" -- a turn i whose tail position n - i + 1 still holds a top element x:
// x is taken off its leaf and re-inserted from the root (it sinks past strictly greater elements along the path of
// larger children and stops at `h`).  The wave-serial replay in beam_exact.hip handles a few dozen tail candidates; a
// wide beam over a 20k-word lexicon has hundreds per frame (a tenth of the top 4000 sit on tail leaves), and replaying
// them one after the other costs more than the loop itself.  This file resolves them together.
//
// THE STATIC MODEL.  Give every top element a chain of incarnations: its heap position after heapify, and one more
// position per event (the landing h).  With the last incarnation of every element in place FROM THE START and the
// earlier ones taken out, the event-free closed form describes the real loop at all times that matter:
//   * the element picked off its leaf never moved before its turn, so nothing ever waited for it (taking it out
//     changes nobody's moves), and
//   * the landed element x at h is worse than everything that passes through h before it lands, so it delays nobody
//     before its turn (putting it there early changes nobody's moves before that turn).
// In the closed form the moves are a table: T_d(e) = the turn in which e leaves the depth-d ancestor of its position,
// T_0(e) = its own extraction turn = 1 + number of elements before it in the order, and
//      T_{d+1}(e) = T_d(the element just before e among those below the same depth-d ancestor)
// (a position passes its subtree's elements up in order, one each time its own occupant leaves).  So ONE SWEEP over the
// list in extraction order gives every move: level by level the elements that lie deeper than d are partitioned
// stably by their next path bit -- all left-goers first, as in a wavelet matrix: the groups (one per depth-d ancestor)
// stay contiguous and sorted -- and take the T of the entry in front of them.  From the table:
//   * an element on the tail position of turn i (depth D) is an event iff T_D >= i (it has not left its leaf), and
//   * its landing: the elements that move in turn i from depth 1, 2, ... are the hole's path; x sinks past them while
//     they are strictly better and takes the position the last of them left.
// An event changes the model (x moves to h), which can change later candidates' answers -- in the real frames of the
// C4 task a quarter of the events see an earlier one.  So the sweep is ITERATED: the incarnations that were picked
// ride along as PROBES (entries that learn when they would have left their leaf but delay nobody: whoever stands
// behind a probe looks through it), every round re-derives every chain from the table, and the rounds stop when no
// chain changes.  A landed entry born in turn b is looked through while the value behind it is < b, which keeps a
// tentative landing from disturbing earlier turns; with that, a self-consistent state is the real loop's (induction
// over the turns: the earliest wrong belief would be re-derived from correct earlier ones).  Real frames converge in 3-6
// rounds.  Inside a group of EQUAL scores the members come out by pre-order among those present; a member that lands
// in turn b is present from turn b + 1 (the schedule in sweep_order_group()).
// tools/prune_lab2.py restates all of this in numpy and checks it against the sequential loop (3 600 tie-heavy random
// arrays, frames of the C4 task); tests/test_prune_order.py does it for this code through jamd_beam_prune_order().
//
// Cost: a round is ~14 level passes over <= 5 000 entries by the whole workgroup (sweep_replay(): each lane takes a run of
// consecutive entries, ~35 instructions an entry and level); whatever the sweep cannot hold
// (more than kSwEvMax events, a tie group with events and more than kSwGroup entries, no convergence) falls back to the
// pipelined extraction loop.
#pragma once

constexpr int kSwEvMax = 1024;           // events (probe entries) held
constexpr int kSwDepth = 16;             // path row: one slot per depth (the sweep holds heaps of < 2^16 tokens: depths 0 .. 15), two 16-byte words
constexpr int kSwChain = 8;              // events per element (an element can bounce from tail leaf to tail leaf: five in real frames)
constexpr int kSwChainRec = 12;          // words of an element's new chain in the global scratch: landings, then the first position
constexpr int kSwGroup = 48;             // entries of a tie group that holds events (ordered by one thread)
constexpr int kSwRounds = 24;
constexpr int kSwPerThread = 8;          // list entries per thread in a pass

typedef JAMD_LDS unsigned short lds_u16;
// the sweep's global scratch is addressed as GLOBAL memory (a generic pointer makes flat_store / flat_load, which also count
// against lgkmcnt: the LDS-only barrier of a level pass would then wait for the path rows' stores after all)
typedef __attribute__((address_space(1))) unsigned short glb_u16;
typedef __attribute__((address_space(1))) unsigned glb_u32;
typedef __attribute__((address_space(1))) u32x4 glb_u32x4;
constexpr unsigned kSwPos = 0xffffu;     // evp[]: heap position (the sweep holds heaps of < 2^16 tokens); bits 16..29 of a landed element's
constexpr int kSwBirthSh = 16;           // word: the turn in which it took that position (13 bits), then that turn's kSwTC flag
constexpr unsigned kSwProbe = 0x80000000u, kSwLanded = 0x40000000u;
// a list entry: entry number (13 bits) | landed << 14 | probe << 15 | T << 16
constexpr unsigned kSwX = 0x1fffu, kSwXLanded = 0x4000u, kSwXProbe = 0x8000u;
constexpr unsigned kSwTV = 0x1fffu;     // the turn itself in the T field (T <= M <= kSwX)
constexpr unsigned kSwTC = 0x8000u;     // T field: this turn is a candidate turn (its tail position holds a top element) up to the limit
// Inside the level passes an entry is TWO slots that move together: a 32-bit word (entry number | landed << 14 | probe << 15 |
// ROUTE << 16) and a 16-bit T field.  ROUTE = the path bits of the entry's position BELOW the level at hand, left-aligned, then a
// terminating 1: 0x8000 = the entry lies at this level's depth, < 0x8000 = it goes left, > 0x8000 = right; the next level's route
// is one shift.  (So a level needs neither the position nor its depth: heaps of < 2^16 tokens.)

// What sort_token_downward() needs from the sweep beside the chains (all in LDS, behind the sweep's own image):
struct SweepDown {
  lds_u32 *fd;                   // [cnt + 1] per turn: (depth << 21 | position) of the deepest position an element left = where the hole
                                 //           leaves the extracted region (0: the root)
  lds_u32 *posend;               // [kSwLeft] position after the last turn of the elements that are not extracted (ties on the cut), by rank - (nB - kSwLeft)
  lds_u32 *evbits;               // [(cnt + 31) / 32 + 1] turns that are events (bit i - 1)
  int want_order;                // 1: svid[] receives the extraction order as well (the caller wants the whole array: residual heap AND extracted part)
};
constexpr int kSwLeft = 512;
__host__ __device__ inline int sweep_down_bytes(int cnt) { return 4 * (cnt + 1) + 4 * kSwLeft + 4 * ((cnt + 31) / 32 + 1) + 48; }

struct SweepMem {
  lds_u32 *evp;                  // [M] position | flags, per entry (x < nB: the element's current incarnation; nB + c: the probe of event c)
  lds_u16 *gid;                  // [nB] tie group: 0x8000 | length at the group's first rank, else that rank
  lds_u16 *ep;                   // [nB + 1] events of the elements in front (the event table is sorted by element)
  lds_u32 *ent[2];               // [M] ent[1]: the list of a level, entry | route << 16, rewritten in place (a pass holds its entries in
                                 //     registers between its two barriers); ent[0]: level 0 as it is built (entry | T << 16), scratch
  lds_u16 *tl;                   // [M] the T fields of the list
  lds_u32 *evq[2], *evh[2];      // [kSwEvMax] event table, double buffered: tail position left, landing
  lds_u16 *evel[2];              // [kSwEvMax] element
  lds_u32 *tailmask;
  lds_u16 *TDx;                  // [M] T at the entry's own depth (lies over ent[0], which the level passes do not use)
  lds_u32 *wsum;                 // [2][NT / 64] wave sums of a level's scan, double buffered
  const lds_u32 *cur_evq;        // the event table in use (a landed element's birth = the turn of its last event)
  int n;
  glb_u16 *path;                 // global [k + 1][kSwDepth] who moves in a candidate turn, by turn and depth
  glb_u32 *ids;                  // global [nB] token ids
  glb_u32 *chain;                // global [nB][kSwChainRec] new chain of an element: its landings, word kSwChain = its first position
  glb_u16 *cand;                 // global [nB] the elements that start on a tail position (only they can have a chain)
};

// LDS and global scratch the sweep needs for a top list of nB entries
__host__ __device__ inline int sweep_lds_bytes(int nB, int k, int evmax) {
  const int M = nB + evmax;
  return 4 * M + 2 * M + 2 * nB + 2 * (nB + 2) + 8 * M + 2 * (4 + 4 + 2) * evmax + 4 * ((k + 31) / 32 + 4) + 512;
}
// the most events (a multiple of 64, at most kSwEvMax) whose image fits `avail` bytes of LDS; 0 = none
__host__ __device__ inline int sweep_pick_evmax(int nB, int k, int avail) {
  int ev = kSwEvMax;
  while (ev >= 64 && sweep_lds_bytes(nB, k, ev) > avail) ev -= 64;
  return ev >= 64 ? ev : 0;
}
__host__ __device__ inline size_t sweep_global_bytes(int b_cap) {
  return 2 * (size_t)(b_cap + 2) * kSwDepth + 4 * (size_t)b_cap + 4 * (size_t)kSwChainRec * (size_t)b_cap + 2 * (size_t)(b_cap + 8) + 64;
}

__device__ __forceinline__ int sw_depth(unsigned p) { return 31 - __clz((int)p); }
// token ids of the top list in the sweep's global scratch (rank order of the list handed to sweep_replay())
__device__ __forceinline__ unsigned *sweep_ids(unsigned char *gs) { return reinterpret_cast<unsigned *>(gs); }
// workgroup barrier that orders LDS traffic only: the global stores of a level pass (who moves when: consumed after the
// whole sweep) stay in flight instead of being waited for at every barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// turn in which landed element x (kSwLanded set) took its present position
__device__ __forceinline__ int sw_birth(const SweepMem &m, unsigned x) { return m.n - (int)m.cur_evq[(int)m.ep[x + 1] - 1] + 1; }

// T field of a turn as it enters the table: bit kSwTC marks a candidate turn (its tail position holds a top element) up to `limit`
__device__ __forceinline__ unsigned sweep_tflag(const lds_u32 *tail, int limit, unsigned T) {
  if (T - 1u < (unsigned)limit) { const unsigned tw = tail[(T - 1u) >> 5]; if ((tw >> ((T - 1u) & 31u)) & 1u) return T | kSwTC; }
  return T;
}
// T field (turn | flag) of the turn in which landed element x took its present position, out of its evp[] word
__device__ __forceinline__ unsigned sweep_birth_field(unsigned e) { const unsigned b = e >> kSwBirthSh; return (b & kSwTV) | ((b << 2) & kSwTC); }
// route of heap position p (depth <= 15) at level 0
__device__ __forceinline__ unsigned sweep_route(unsigned p) { return (((p << 1) | 1u) << (15 - sw_depth(p))) & 0xffffu; }
// inclusive prefix sums over the lanes of a wave / over each row of 16 lanes, on the DPP path (no LDS round trips)
__device__ __forceinline__ unsigned dpp_row_scan(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  return v;
}
__device__ __forceinline__ unsigned dpp_wave_scan(unsigned v) {
  v = dpp_row_scan(v);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}

// block-wide exclusive scan with max (one unsigned per thread, identity 0)
template <int NT>
__device__ __forceinline__ unsigned block_excl_scan_max(XShared &sh, unsigned v) {
  const int tx = tid_now(), lane = tx & 63, wv = tx >> 6;
  unsigned incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off, 64);
    if (lane >= off) incl = incl > o ? incl : o;
  }
  if (lane == 63) sh.wsum[wv] = incl;
  __syncthreads();
  unsigned base = 0;
  for (int w = 0; w < wv; w++) base = base > sh.wsum[w] ? base : sh.wsum[w];
  unsigned ex = __shfl_up(incl, 1, 64);
  if (lane == 0) ex = 0;
  __syncthreads();
  return base > ex ? base : ex;
}

// order key of an entry inside its tie group: pre-order of its position, then the entry number
__device__ __forceinline__ unsigned long long sw_key(const SweepMem &m, unsigned x) {
  return ((unsigned long long)prekey(m.evp[x] & kSwPos) << 16) | x;
}

// A tie group that holds events, ordered by ONE thread: its nm members' current incarnations and the probes of their
// events [c0, c1), into dst[0 .. cnt) as entry | T_0 << 16.  Pre-order among the entries; the members come out in that
// order among those PRESENT (a member landed in turn b is present from turn b + 1), a probe stays in front of the member
// that follows it in pre-order.  `tmp` = cnt words of scratch.  Returns false when the group is too large.
__device__ __forceinline__ bool sweep_order_group(const SweepMem &m, int nB, int g0, int nm, int c0, int c1, lds_u32 *dst, lds_u32 *tmp, int k, lds_i32 *svid) {
  const int cnt = nm + (c1 - c0);
  if (cnt > kSwGroup) return false;
  for (int i = 0; i < cnt; i++) {                                      // insertion sort by (pre-order, entry)
    const unsigned x = i < nm ? (unsigned)(g0 + i) : (unsigned)(nB + c0 + (i - nm));
    const unsigned long long kx = sw_key(m, x);
    int j = i;
    while (j > 0 && sw_key(m, dst[j - 1]) > kx) { dst[j] = dst[j - 1]; j--; }
    dst[j] = x;
  }
  const int tstart = g0 + 1;                                           // turn of the group's first member
  bool sched = false;
  for (int i = 0; i < cnt; i++) {
    const unsigned x = dst[i];
    if (x < (unsigned)nB && (m.evp[x] & kSwLanded) && sw_birth(m, x) >= tstart) sched = true;
  }
  if (sched) {
    // turn by turn: the first member in pre-order that is present and not out yet
    unsigned long long out = 0ull;
    for (int i = 0; i < cnt; i++) tmp[i] = 0u;                         // tmp[i] = turn of dst[i] (members)
    int t = tstart;
    for (int r = 0; r < nm; r++, t++) {
      int pick = -1, early = -1;
      for (int i = 0; i < cnt; i++) {
        const unsigned x = dst[i];
        if (x >= (unsigned)nB || ((out >> i) & 1ull)) continue;
        const int b = (m.evp[x] & kSwLanded) ? sw_birth(m, x) : 0;
        if (b < t) { pick = i; break; }
        if (early < 0 || b < ((m.evp[dst[early]] & kSwLanded) ? sw_birth(m, dst[early]) : 0)) early = i;
      }
      if (pick < 0) pick = early;                                       // (not in a consistent state)
      out |= 1ull << pick;
      tmp[pick] = (unsigned)t;
    }
    // new order: members by turn, every probe in front of the member that followed it in pre-order
    unsigned follow = 0u;                                              // turn of the next member in pre-order, 0 = none
    for (int i = cnt - 1; i >= 0; i--) {
      if (dst[i] < (unsigned)nB) follow = tmp[i]; else tmp[i] = follow ? follow : 0xffffu;   // probe: the turn it precedes
    }
    // stable insertion sort by (turn, probes first, pre-order): cnt <= kSwGroup
    for (int i = 0; i < cnt; i++) {
      const unsigned turn = tmp[i] == 0xffffu ? (unsigned)(tstart + nm) : tmp[i];
      tmp[i] = (turn << 16) | (dst[i] >= (unsigned)nB ? 0u : 0x8000u) | (unsigned)i;
    }
    for (int i = 1; i < cnt; i++) {
      const unsigned kx = tmp[i], x = dst[i];
      int j = i;
      while (j > 0 && tmp[j - 1] > kx) { tmp[j] = tmp[j - 1]; dst[j] = dst[j - 1]; j--; }
      tmp[j] = kx; dst[j] = x;
    }
    for (int i = 0; i < cnt; i++) dst[i] |= tmp[i] & 0xffff0000u;
  } else {
    int t = tstart;
    for (int i = 0; i < cnt; i++) {
      const unsigned x = dst[i];
      dst[i] = x | ((unsigned)t << 16);
      if (x < (unsigned)nB) t++;
    }
  }
  for (int i = 0; i < cnt; i++) {                                      // the order of the next frame (last round's values stand)
    const unsigned w = dst[i], x = w & 0xffffu;
    const int t = (int)(w >> 16);
    if (x < (unsigned)nB && t <= k) svid[k - t] = (int)m.ids[x];
    dst[i] = w | (x >= (unsigned)nB ? kSwXProbe : ((m.evp[x] & kSwLanded) ? kSwXLanded : 0u));
  }
  return true;
}

// The replay itself.  In: the top list sorted by (score, pre-order) -- pm.compR / vposR / idR of nB entries --, the tail
// mask of the tail positions that hold one (bit b <-> turn b + 1), ilast = the last turn at which a TIED element can sit
// on the tail position.  `region` = LDS the sweep may lay out afresh (the lists are read into registers first), gs =
// global scratch of sweep_global_bytes().  Out: svid[0..k) = token ids in the visiting order of the next frame.
// Returns false when it gives up (the caller runs the extraction loop).  Whole workgroup.
template <int NT>
__device__ __noinline__ bool sweep_replay(XShared &sh, unsigned char JAMD_LDS *region, int region_bytes, unsigned char *gs,
                                          const lds_u64 *compR, const lds_u32 *vposR, const lds_u32 *idR, lds_u32 *tailmask,
                                          int nB, int n, int k, int ilast, lds_i32 *svid, int evmax, const SweepDown *down) {
  const int tid = tid_now();
  const unsigned long long clk0 = wall_clock64();
  unsigned long long clk = clk0;
#define SWTICK(i) do { if (tid == 0) { const unsigned long long c_ = wall_clock64(); sh.sw_prof[i] += (int)(c_ - clk); clk = c_; } } while (0)
#ifdef JAMD_SWEEP_LEVEL_TICKS                                          // development: the clock of thread JAMD_SWEEP_LEVEL_TICKS's wave inside a level pass (slots 6, 7, and 4, 5 borrowed)
#define SWTICK2(i) do { if (tid == JAMD_SWEEP_LEVEL_TICKS) { const unsigned long long c_ = wall_clock64(); sh.sw_prof[i] += (int)(c_ - clk2); clk2 = c_; } } while (0)
#else
#define SWTICK2(i) do { } while (0)
#endif
  const int M = nB + evmax;
  const int nwords = (k + 31) / 32 + 1;
  if (n >= 0x10000 || nB > kSwPerThread * NT || M > (int)kSwX || nB >= 0x8000 || sweep_lds_bytes(nB, k, evmax) > region_bytes || 4 * (2 * nB + nwords) > 4 * kSwChainRec * nB) return false;
  // ---- the lists leave the region through the global scratch, then it is laid out afresh
  SweepMem m;
  m.n = n;
  m.ids = (glb_u32 *)sweep_ids(gs);
  m.chain = m.ids + ((nB + 3) & ~3);
  m.cand = (glb_u16 *)(m.chain + (size_t)kSwChainRec * nB);
  m.path = m.cand + ((nB + 7) & ~7);                                   // (16-byte aligned: rows are read as three 16-byte words)
  glb_u32 *const stage = m.chain;      // [nB] positions, [nB] score bits, the tail mask
  for (int r = tid; r < nB; r += NT) { stage[r] = vposR[r]; stage[nB + r] = (unsigned)(compR[r] >> 32); m.ids[r] = idR[r]; }
  for (int w = tid; w < nwords; w += NT) stage[2 * nB + w] = tailmask[w];
  __syncthreads();
  {
    unsigned char JAMD_LDS *at = region;
    auto take = [&](int bytes) { unsigned char JAMD_LDS *p = at; at += (bytes + 15) & ~15; return p; };
    m.evp = (lds_u32 *)take(4 * M);
    m.ent[0] = (lds_u32 *)take(4 * M);
    m.ent[1] = (lds_u32 *)take(4 * M);
    m.tl = (lds_u16 *)take(2 * M);
    m.TDx = (lds_u16 *)m.ent[0];
    m.gid = (lds_u16 *)take(2 * nB);
    m.ep = (lds_u16 *)take(2 * (nB + 2));
    for (int b = 0; b < 2; b++) { m.evq[b] = (lds_u32 *)take(4 * evmax); m.evh[b] = (lds_u32 *)take(4 * evmax); m.evel[b] = (lds_u16 *)take(2 * evmax); }
    m.tailmask = (lds_u32 *)take(4 * (nwords + 1));
    m.wsum = (lds_u32 *)take(4 * 2 * (NT / 64));
  }
  lds_u32 *score = m.ent[1];                                           // (until the groups are known)
  for (int r = tid; r < nB; r += NT) { m.evp[r] = stage[r]; score[r] = stage[nB + r]; }
  for (int w = tid; w < nwords; w += NT) m.tailmask[w] = stage[2 * nB + w];
  if (tid == 0) m.tailmask[nwords] = 0u;
  __syncthreads();
  // ---- tie groups: gid[r] = first rank of r's group; the first rank carries 0x8000 | length
  {
    const int C = (nB + NT - 1) / NT;
    const int lo = tid * C, hi = min(nB, lo + C);
    unsigned last = 0u;                                                // (first rank + 1) of the latest group start in the chunk
    for (int r = lo; r < hi; r++) if (r == 0 || score[r - 1] != score[r]) last = (unsigned)r + 1u;
    const unsigned before = block_excl_scan_max<NT>(sh, last);
    unsigned cur = before;
    for (int r = lo; r < hi; r++) {
      if (r == 0 || score[r - 1] != score[r]) cur = (unsigned)r + 1u;
      m.gid[r] = (unsigned short)(cur - 1u);
    }
    __syncthreads();
    lds_u32 *glen = m.ent[0];
    for (int r = tid; r < nB; r += NT) glen[r] = 0u;
    __syncthreads();
    for (int r = tid; r < nB; r += NT) atomicMax((unsigned *)&glen[m.gid[r]], (unsigned)(r - (int)m.gid[r] + 1));
    __syncthreads();
    for (int r = tid; r < nB; r += NT) if ((int)m.gid[r] == r) m.gid[r] = (unsigned short)(0x8000u | glen[r]);
    __syncthreads();
  }
  auto group_of = [&](int r, int &g0, int &len) {
    const unsigned g = m.gid[r];
    g0 = (g & 0x8000u) ? r : (int)g;
    len = (int)(m.gid[g0] & 0x7fffu);
  };
  if (tid == 0) { sh.sw_nev = 0; sh.sw_limit = ilast; sh.sw_fail = 0; sh.sw_ncl = 0; }
  for (int r = tid; r <= nB + 1; r += NT) m.ep[r] = 0;                  // (kept up to date where the event table is rebuilt)
  __syncthreads();
  // Only an element that STARTS on a tail position can have a chain (a tenth of the list): they are gathered once, so that
  // the chain walk of a round runs on a few full waves instead of on every wave with a tenth of its lanes.
  for (int r0 = 0; r0 < nB; r0 += NT) {
    const int r = r0 + tid;
    const bool has = r < nB && (m.evp[r] & kSwPos) >= (unsigned)(n - k + 1);
    const int slot = wave_alloc(&sh.sw_ncl, has);
    if (has) m.cand[slot] = (unsigned short)r;
  }
  __syncthreads();
  SWTICK(0);
  int cur = 0;                                                         // event table in use
  int round = 0;
  for (;;) {
    round++;
    if (round > kSwRounds) return false;
    const int nev = uni(sh.sw_nev), limit = uni(sh.sw_limit);
    const int L0 = nB + nev;
    lds_u32 *const evq = m.evq[cur], *const evh = m.evh[cur];
    m.cur_evq = evq;
    // ---- the path rows of the candidate turns cleared (ep[r] = events of the elements in front of r: from the rebuild below)
    if (down) {
      for (int i = tid; i <= k; i += NT) down->fd[i] = 0u;
      for (int i = tid; i < kSwLeft; i += NT) down->posend[i] = 0u;
    }
    for (int w = tid; w < nwords; w += NT) {                             // (a row is written and read in the same round, for the turns flagged below)
      unsigned tw = m.tailmask[w];
      while (tw) {
        const int T = 32 * w + __ffs((int)tw);
        tw &= tw - 1u;
        if (T > limit) break;
        glb_u32x4 *row = (glb_u32x4 *)(m.path + (size_t)T * kSwDepth);
        const u32x4 ff = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        row[0] = ff; row[1] = ff;
      }
    }
    SWTICK(1);
    // ---- level 0: the entries in extraction order, as the passes want them: entry | route in the list, the T fields beside
    // it, a landed entry's birth with its flag in its evp[] word.  A group without events keeps the order of the sorted list.
    const lds_u16 *const l_ep = m.ep;                                   // (locals: `m` itself may live in scratch memory)
    const lds_u32 *const l_evq = evq, *const l_evp = m.evp, *const l_tail = m.tailmask;
    lds_u16 *const l_TDx = m.TDx;
    lds_u32 *const l_evpw = m.evp;
    glb_u16 *const l_path = m.path;
    lds_u32 *const l_wsum = m.wsum, *const l_ent0 = m.ent[0], *const l_ent1 = m.ent[1];
    lds_u16 *const l_t1 = m.tl;
    lds_u32 *const l_fd = down ? down->fd : nullptr, *const l_posend = down ? down->posend : nullptr;
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int NW = NT / 64;
    // A turn's flag (bit kSwTC of the T field: "this is a candidate turn up to `limit`": its path row is wanted) is looked
    // up ONCE, where the value enters the table, and rides along with it.
    auto tflag = [&](unsigned T) -> unsigned { return sweep_tflag(l_tail, limit, T); };
    for (int r = tid; r < nB; r += NT) {
      int g0, len;
      group_of(r, g0, len);
      const int c0 = l_ep[g0], c1 = l_ep[g0 + len];
      if (c0 == c1) {
        l_ent1[r + c0] = (unsigned)r | (sweep_route(l_evp[r] & kSwPos) << 16);
        l_t1[r + c0] = (unsigned short)tflag((unsigned)(r + 1));
        if ((!down || down->want_order) && r + 1 <= k) svid[k - (r + 1)] = (int)m.ids[r];
      } else if (r == g0) {
        lds_u32 *const dst = l_ent0 + g0 + c0;                          // (ordered as entry | T << 16, scratch in the list's own place)
        if (!sweep_order_group(m, nB, g0, len, c0, c1, dst, l_ent1 + g0 + c0, (down && !down->want_order) ? 0 : k, svid)) sh.sw_fail = 1;
        else {
          for (int i = 0; i < len + (c1 - c0); i++) {
            const unsigned w = dst[i];
            const unsigned x = w & kSwX;
            l_ent1[g0 + c0 + i] = (w & 0xdfffu) | (sweep_route(l_evp[x] & kSwPos) << 16);
            l_t1[g0 + c0 + i] = (unsigned short)tflag(w >> 16);
            if (w & kSwXLanded) {
              const unsigned bf = tflag((unsigned)(n - (int)l_evq[(int)l_ep[x + 1u] - 1] + 1));
              l_evpw[x] = (l_evpw[x] & (kSwPos | kSwProbe | kSwLanded)) | (((bf & kSwTV) | ((bf & kSwTC) >> 2)) << kSwBirthSh);
            }
          }
        }
      }
    }
    __syncthreads();
    if (uni(sh.sw_fail)) return false;
    SWTICK(2);
    // ---- the sweep: level d -> d + 1
    int L = L0;
    unsigned long long clk2 = wall_clock64(); (void)clk2;
    for (int d = 0; L > 0; d++) {
      if (d >= kSwDepth) return false;
      // Lane-runs: a thread takes CH consecutive entries of the list, so the entry in front of all but its first is its own,
      // its place among the left- / right-goers is a count inside the thread plus ONE prefix sum per wave (DPP), and every
      // quantity of an entry comes out of its two slots by a compare (round 6; the round-4 form gave every lane one entry of a
      // 64-entry row and paid ~100 instructions an entry and level for ballots, position look-ups and nested branches).
      // A lane's run is 4 or 8 entries: one or two 16-byte reads of the list, 8-byte reads of the T fields (a run that starts
      // anywhere makes the lanes' 4-byte reads collide in the LDS banks).  The list falls into units of 256 entries; every wave
      // takes one, the first `nb` waves a second one.
      const int nunits = (L + 255) >> 8, nb = max(nunits - NW, 0);
      if (nb > NW) return false;
      const int wvu = uni(wv);
      const int CH = wvu < nb ? 8 : 4;
      const int wavebase = wvu < nb ? wvu * 512 : nb * 512 + (wvu - nb) * 256;
      lds_u32 *const Wa = l_ent1, *const Wb = l_ent1;                  // (in place: every entry is in a register between the two barriers)
      lds_u16 *const Ta = l_t1, *const Tb = l_t1;
      const int base = wavebase + lane * CH;
      const int nval = min(max(L - base, 0), CH);
      const lds_u32 *const wp = Wa + base;
      const lds_u16 *const tp = Ta + base;
      unsigned w[kSwPerThread], f[kSwPerThread];                       // f: the T field an entry takes to the next level
      unsigned cl = 0u, cr = 0u, accw = 0u, acct = 0u;
#pragma unroll
      for (int i = 0; i < kSwPerThread; i++) { w[i] = 0u; f[i] = 0u; }
      const bool active = wavebase < L;                                 // (a wave behind the end of a short list only joins the barriers)
      if (active) {
        unsigned t[kSwPerThread];
        unsigned wq = wp[-1], v = tp[-1];                               // the entry in front of the run
        if (base == 0 || nval == 0) { wq = 0u; v = 0u; }
        {
          typedef JAMD_LDS unsigned long long lds_u64x;
          const u32x4 a0 = *(const lds_v4 *)wp;
          const unsigned long long b0 = *(const lds_u64x *)tp;
          u32x4 a1 = {0u, 0u, 0u, 0u};
          unsigned long long b1 = 0ull;
          if (CH == 8) { a1 = *(const lds_v4 *)(wp + 4); b1 = *(const lds_u64x *)(tp + 4); }
          const unsigned aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
          for (int i = 0; i < kSwPerThread; i++) {
            const unsigned b = (unsigned)((i < 4 ? b0 : b1) >> (16 * (i & 3))) & 0xffffu;
            w[i] = i < nval ? aw[i] : 0u; t[i] = i < nval ? b : 0u;
          }
        }
#pragma unroll
        for (int i = 0; i < kSwPerThread; i++) {
          if (i < CH) {
            cl += (int)w[i] > 0xffff ? 1u : 0u;                         // 0 < route < 0x8000
            cr += w[i] >= 0x80010000u ? 1u : 0u;                        // route > 0x8000
            accw |= w[i]; acct |= t[i];
            if ((w[i] >> 16) == 0x8000u) l_TDx[w[i] & kSwX] = (unsigned short)t[i];   // T at the entry's own depth
          }
        }
        SWTICK2(6);
        if (d >= 1 && __any((int)(acct & kSwTC))) {                     // who moves in a candidate turn
#pragma unroll
          for (int i = 0; i < kSwPerThread; i++) {
            if (i < CH) {
              if ((t[i] & kSwTC) && !(w[i] & kSwXProbe)) l_path[(t[i] & kSwTV) * (unsigned)kSwDepth + (unsigned)d] = (unsigned short)(w[i] & kSwX);
            }
          }
        }
        SWTICK2(0);
        if (l_fd) {                                                    // sorting downward: where the hole leaves the extracted region
#pragma unroll                                                         // (the deepest mover of a turn: entry number here, its position after the last round),
          for (int i = 0; i < kSwPerThread; i++) {                     // and where the elements that stay end up
            if (i < CH) {
              if (w[i] != 0u && !(w[i] & kSwXProbe)) {
                const unsigned T = t[i] & kSwTV, x = w[i] & kSwX;
                if ((int)T <= k) { if (d >= 1) atomicMax((unsigned *)&l_fd[T], ((unsigned)d << 21) | x); }
                else {
                  const int li = (int)x - (nB - kSwLeft);
                  const unsigned vp = l_evp[x] & kSwPos;
                  if (li >= 0) l_posend[li] = vp >> (sw_depth(vp) - d); else sh.sw_fail = 1;
                }
              }
            }
          }
        }
        // What the entry behind reads is the T of the entry in front -- except behind a probe (looked through: it delays
        // nobody) and behind a landed entry born in turn b (looked through while the value in front of it is < b).  The
        // value runs along the thread's entries in registers; only a thread whose run begins behind such an entry walks
        // back through the list.  (The round-4 form had these entries rewrite their slots before a barrier: 2.5 of a
        // level's 4.3 us were that walk and the wait for the slowest wave's.)
        if (__any((int)((accw | wq) & (kSwXProbe | kSwXLanded)))) {
          // (all look-ups first, the two entries in front of the run among them: the walk below is for the thread whose run
          // begins behind two or more such entries)
          unsigned wq2 = wp[-2], tq2 = tp[-2];
          if (base <= 1 || nval == 0) { wq2 = 0u; tq2 = 0u; }
          unsigned bq = 0xffffu, bb[kSwPerThread];                      // (a probe: as if born after every turn)
          if (wq & kSwXLanded) bq = sweep_birth_field(l_evp[wq & kSwX]);
#pragma unroll
          for (int i = 0; i < kSwPerThread; i++) { bb[i] = 0xffffu; if (i < CH) { if (w[i] & kSwXLanded) bb[i] = sweep_birth_field(l_evp[w[i] & kSwX]); } }
          if (wq & (kSwXProbe | kSwXLanded)) {
            unsigned in = tq2;                                          // what stands in front of the entry in front
            if (wq2 & (kSwXProbe | kSwXLanded)) {
              int j = base - 2;
              while (j >= 0 && (Wa[j] & (kSwXProbe | kSwXLanded))) j--;
              in = j >= 0 ? (unsigned)Ta[j] : 0u;
              for (int s = j + 1; s < base - 1; s++) {
                const unsigned ws = Wa[s];
                if (ws & kSwXProbe) continue;
                const unsigned b = sweep_birth_field(l_evp[ws & kSwX]);
                if ((in & kSwTV) < (b & kSwTV)) continue;
                const unsigned ts = Ta[s];
                in = (ts & kSwTV) > (b & kSwTV) ? ts : b;
              }
            }
            const unsigned late = (v & kSwTV) > (bq & kSwTV) ? v : bq;  // (v: the T field of the entry in front)
            v = (in & kSwTV) < (bq & kSwTV) ? in : late;
          }
#pragma unroll
          for (int i = 0; i < kSwPerThread; i++) {
            f[i] = 0u;
            if (i < CH) {
              f[i] = v;
              const unsigned b = bb[i];
              const unsigned late = (t[i] & kSwTV) > (b & kSwTV) ? t[i] : b;
              const unsigned thru = (v & kSwTV) < (b & kSwTV) ? v : late;
              v = (w[i] & (kSwXProbe | kSwXLanded)) ? thru : t[i];
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < kSwPerThread; i++) { f[i] = 0u; if (i < CH) { f[i] = v; v = t[i]; } }
        }
      }
      SWTICK2(1);
      // where this thread's left- and right-goers start: prefix over the wave, then over the waves' totals
      const unsigned c = cl | (cr << 16);
      const unsigned incl = dpp_wave_scan(c);
      lds_u32 *ws = l_wsum + (d & 1) * NW;                              // (one barrier: the totals alternate between two buffers)
      if (lane == 63) ws[wv] = incl;
      lds_barrier();
      SWTICK2(7);
      const unsigned wsc = dpp_row_scan(lane < NW ? ws[lane] : 0u);
      const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)wsc, NW - 1);
      const unsigned wbase = wvu ? (unsigned)__builtin_amdgcn_readlane((int)wsc, wvu - 1) : 0u;
      const int totl = (int)(tot & 0xffffu), totr = (int)(tot >> 16);
      const unsigned ex = wbase + incl - c;
      unsigned el = ex & 0xffffu, er = (unsigned)totl + (ex >> 16);
      if (active) {
#pragma unroll
      for (int i = 0; i < kSwPerThread; i++) {
        if (i < CH) {
          const bool left = (int)w[i] > 0xffff, right = w[i] >= 0x80010000u;
          if (left || right) {
            const unsigned at = left ? el : er;
            Wb[at] = (w[i] & 0xffffu) | ((w[i] & 0x7fff0000u) << 1);
            Tb[at] = (unsigned short)f[i];
          }
          el += left ? 1u : 0u; er += right ? 1u : 0u;
        }
      }
      }
      SWTICK2(4);
      lds_barrier();
      SWTICK2(5);
      L = totl + totr;
    }
    __syncthreads();                                                   // (the moves written to global memory)
    SWTICK(3);
    // ---- every chain again from the table
    if (tid == 0) { sh.sw_changed = 0; }
    lds_u16 *const ncnt = (lds_u16 *)m.ent[1];                         // new events per element (the list is done with; ent[0] holds TDx[])
    lds_u16 *const nep = ncnt + ((nB + 9) & ~7);                       // ep[] of the next round
    for (int r = tid; r <= nB; r += NT) ncnt[r] = 0;
    __syncthreads();
    const int ncl = uni(sh.sw_ncl);
    for (int ci = tid; ci < ncl; ci += NT) {
      const int r = (int)m.cand[ci];
      const int c0 = m.ep[r], oc = (int)m.ep[r + 1] - c0;
      const unsigned q0 = oc ? evq[c0] : (m.evp[r] & kSwPos);
      int g0, len;
      group_of(r, g0, len);
      unsigned nh[kSwChain];
      int nn = 0;
      unsigned vp = q0;
      bool changed = false;
      for (int t = 0; t < kSwChain; t++) {
        if (vp < (unsigned)(n - k + 1)) break;
        const int turn = n - (int)vp + 1;
        if (turn > limit) break;
        if (t > oc || (t > 0 && nh[t - 1] != evh[c0 + t - 1])) break;   // this incarnation was not in the sweep: next round
        const unsigned x = t < oc ? (unsigned)(nB + c0 + t) : (unsigned)r;
        if ((int)(m.TDx[x] & kSwTV) < turn) break;                      // it left the leaf before its turn
        // landing: past the elements that move in this turn while they are strictly better
        const unsigned tw = m.tailmask[(turn - 1) >> 5], bit = 1u << ((turn - 1) & 31);
        if (!(tw & bit)) { sh.sw_fail = 1; break; }
        const glb_u32x4 *row = (const glb_u32x4 *)(m.path + (size_t)turn * kSwDepth);
        const u32x4 r0 = row[0], r1 = row[1];
        const unsigned rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        unsigned h = 1u;
        bool stop = false;
#pragma unroll
        for (int dd = 1; dd < kSwDepth; dd++) {
          const unsigned y = (rw[dd >> 1] >> (16 * (dd & 1))) & 0xffffu;
          if (stop || y == 0xffffu || y == (unsigned)r) { stop = true; continue; }
          int yg, yl;
          group_of((int)y, yg, yl);
          if (!(yg < g0)) { stop = true; continue; }
          const unsigned yv = m.evp[y] & kSwPos;
          h = yv >> (sw_depth(yv) - dd);
        }
        nh[nn++] = h;
        vp = h;
      }
      if (nn == kSwChain && vp >= (unsigned)(n - k + 1) && n - (int)vp + 1 <= limit) sh.sw_fail = 1;   // one more event: not held
      changed = nn != oc;
      for (int t = 0; t < nn && t < oc; t++) changed |= nh[t] != evh[c0 + t];
      if (len > 1) {
        for (int t = 0; t < nn; t++) {
          if (nh[t] >= (unsigned)(n - k + 1) && n - (int)nh[t] + 1 > limit) { atomicMax(&sh.sw_limit, n - (int)nh[t] + 1); changed = true; }
        }
      }
      for (int t = 0; t < nn; t++) {
        if (nh[t] >= (unsigned)(n - k + 1)) { const int b = n - (int)nh[t]; atomicOr((unsigned *)&m.tailmask[b >> 5], 1u << (b & 31)); }
      }
      if (changed) sh.sw_changed = 1;
      ncnt[r] = (unsigned short)nn;
      {
        glb_u32 *rec = m.chain + (size_t)r * kSwChainRec;
        for (int t = 0; t < nn; t++) rec[t] = nh[t];
        rec[kSwChain] = q0;
      }
    }
    __syncthreads();
    SWTICK(4);
    if (uni(sh.sw_fail)) return false;
    if (!uni(sh.sw_changed)) break;
    // ---- the new event table (sorted by element), the entries' positions
    {
      const int C = (nB + 1 + NT - 1) / NT;
      const int lo = tid * C, hi = min(nB + 1, lo + C);
      int tot = 0;
      for (int r = lo; r < hi; r++) tot += (int)ncnt[r];
      int ex = block_excl_scan<NT>(sh, tot);
      const int total = uni(sh.scan_total);
      if (total > evmax) return false;
      lds_u32 *const nq = m.evq[cur ^ 1], *const nhh = m.evh[cur ^ 1];
      lds_u16 *const nel = m.evel[cur ^ 1];
      for (int r = lo; r < hi && r < nB; r++) {
        const int nn = (int)ncnt[r];
        const int oc = (int)m.ep[r + 1] - (int)m.ep[r];
        nep[r] = (unsigned short)ex;
        if (nn == 0 && oc == 0) continue;
        const glb_u32 *rec = m.chain + (size_t)r * kSwChainRec;
        unsigned q = rec[kSwChain];
        for (int t = 0; t < nn; t++) { const unsigned h = rec[t]; nq[ex + t] = q; nhh[ex + t] = h; nel[ex + t] = (unsigned short)r; q = h; }
        m.evp[r] = q | (nn ? kSwLanded : 0u);
        ex += nn;
      }
      __syncthreads();
      for (int r = tid; r < nB; r += NT) m.ep[r] = nep[r];
      if (tid == 0) m.ep[nB] = (unsigned short)total;
      for (int c = tid; c < total; c += NT) m.evp[nB + c] = nq[c] | kSwProbe;
      if (tid == 0) sh.sw_nev = total;
      cur ^= 1;
      __syncthreads();
    }
    SWTICK(5);
  }
  if (down) {
    const int nev = uni(sh.sw_nev);
    for (int i = tid; i < (k + 31) / 32 + 1; i += NT) down->evbits[i] = 0u;
    __syncthreads();
    for (int c = tid; c < nev; c += NT) { const int b = n - (int)m.evq[cur][c]; atomicOr((unsigned *)&down->evbits[b >> 5], 1u << (b & 31)); }
    for (int i = tid; i <= k; i += NT) {                                 // fd[]: the deepest mover of a turn -> the position it left
      const unsigned w = down->fd[i];
      if (w) { const unsigned dd = w >> 21, vp = m.evp[w & kSwX] & kSwPos; down->fd[i] = (dd << 21) | (vp >> (sw_depth(vp) - (int)dd)); }
    }
    if (uni(sh.sw_fail)) return false;
  }
  if (tid == 0) { sh.sw_info = round; sh.sw_ticks = (int)(wall_clock64() - clk0); sh.sw_nev_out = sh.sw_nev; }   // diagnostic: jamd_beam_prune_info()
  __syncthreads();
  return true;
#undef SWTICK
#undef SWTICK2
}

// sort_token_downward() (beam.c:1414-1457), the part the sweep does not give: the residual heap.  Every turn i takes the
// tail element s = H[n - i + 1]; when s is not one of the extracted elements it sinks from the position where the hole
// left the extracted region (fd[i]) through the elements that stay -- a few levels at most, the extracted region covers
// the top of the heap.  The sifts of different turns touch the same positions only when one starts below the other
// (the lower one is the earlier turn: the extracted region shrinks from below) or when a sift ends on the tail position a
// later turn reads; so every sift waits for at most three earlier ones -- the turns that freed its start's two children
// and the latest turn that freed an ancestor of its tail position -- and otherwise all of them run at once, one lane
// each, in windows of NT turns.  P = the heap as heapify left it (global copy, loaded to LDS here).
// MINHEAP = the downward sort; false = the same replay for sort_token_upward() (:1342-1383: the k BEST are extracted from a
// max-heap, the comparisons mirrored) -- wanted when the caller needs tindex[] whole (the multipath frame's mid-frame sort).
// Out: svid[0..k) = the token ids at heap positions 1..k after the last turn (= tindex[0..k), :1512-1514); or, arr_full
// given, arr_full[0..k) instead (global; k may exceed what svid[] holds) and svid[] untouched.
template <int NT, bool MINHEAP>
__device__ __noinline__ bool down_finish(XShared &sh, unsigned char JAMD_LDS *region, int region_bytes, const unsigned long long *Pg,
                                         int n, int k, SweepDown dn, const unsigned *ids, int nB, lds_i32 *svid, int *arr_full) {
  const int tid = tid_now();
  const int R = n - k;
  // LDS image, 8 bytes a token: the score bits by heap-time position (read only), the heap as 16-bit heap-time positions
  // (what the sifts move), the strip, one byte a turn
  const int o_p16 = (4 * (n + 2) + 15) & ~15, o_strip = o_p16 + ((2 * (n + 2) + 15) & ~15), o_done = o_strip + ((2 * (n + 2) + 15) & ~15);
  if (o_done + R + 16 > region_bytes || n >= 0xffff) return false;
  unsigned long long dclk = wall_clock64();
#define DFTICK(i) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); sh.df_prof[i] += (int)(n_ - dclk); dclk = n_; } } while (0)
  const lds_u32 *S = (const lds_u32 *)region;
  volatile lds_u16 *P = (volatile lds_u16 *)(region + o_p16);
  lds_u16 *strip = (lds_u16 *)(region + o_strip);
  volatile JAMD_LDS unsigned char *done = (volatile JAMD_LDS unsigned char *)(region + o_done);
  for (int p = tid; p <= n; p += NT) { ((lds_u32 *)region)[p] = p ? (unsigned)(Pg[p] >> 32) : 0u; P[p] = (unsigned short)p; strip[p] = 0xffffu; }
  for (int i = tid; i <= R; i += NT) done[i] = i == 0 ? 1 : 0;
  __syncthreads();
  auto is_event = [&](int i) { return (dn.evbits[(i - 1) >> 5] >> ((i - 1) & 31)) & 1u; };
  auto start_of = [&](int i) { const unsigned w = dn.fd[i]; return w ? (int)(w & 0x1fffffu) : 1; };
  for (int i = 1 + tid; i <= R; i += NT) if (!is_event(i)) strip[start_of(i)] = (unsigned short)i;
  __syncthreads();
  DFTICK(0);
  for (int base = 0; base < R; base += NT) {
    const int i = base + tid + 1;
    bool mine = i <= R && !is_event(i);
    int f = 1, d1 = 0, d2 = 0, d3 = 0;
    const int q = n - i + 1, m = n - i;
    if (mine) {
      f = start_of(i);
      if (2 * f <= n) { const int s = strip[2 * f]; if (s < i) d1 = s; }
      if (2 * f + 1 <= n) { const int s = strip[2 * f + 1]; if (s < i) d2 = s; }
      for (int a = q; a >= 1; a >>= 1) {                                  // the latest earlier turn that freed the tail position or one above it
        const int s = strip[a];
        if (s == 0xffff) continue;
        if (s < i) d3 = s; else break;
      }
    }
    if (base == 0) DFTICK(1);
    int guard = 0;
    while (__any(mine)) {
      if (mine && done[d1] && done[d2] && done[d3]) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned short s = P[q];
        const unsigned sv = S[s];
        int p = f, child;
        while ((child = 2 * p) <= m) {
          // both children at once: two LDS round trips a level instead of four (the slot behind the heap's end is read and ignored)
          unsigned short c = P[child];
          const unsigned short c2 = P[child + 1];
          unsigned cv = S[c];
          const unsigned cv2 = S[c2];
          if (child < m && (MINHEAP ? (cv > cv2) : (cv < cv2))) { child++; c = c2; cv = cv2; }
          if (MINHEAP ? (sv <= cv) : (sv >= cv)) break;
          P[p] = c;
          p = child;
        }
        P[p] = s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        done[i] = 1;
        mine = false;
      }
      if (++guard > (1 << 22)) { sh.sw_fail = 1; break; }                  // (a dependency that never resolves: not in a consistent state)
    }
    __syncthreads();
    if (uni(sh.sw_fail)) return false;
  }
  DFTICK(2);
  if (arr_full) { for (int j = tid; j < k; j += NT) arr_full[j] = (int)(unsigned)Pg[P[1 + j]]; }
  else { for (int j = tid; j < k; j += NT) svid[j] = (int)(unsigned)Pg[P[1 + j]]; }
  __syncthreads();
  for (int li = tid; li < kSwLeft; li += NT) {                              // the elements of the list that were not extracted (ties on the cut)
    const unsigned pe = dn.posend[li];
    if (pe) {
      if (pe > (unsigned)k) sh.sw_fail = 1;
      else if (arr_full) arr_full[pe - 1] = (int)ids[nB - kSwLeft + li];
      else svid[pe - 1] = (int)ids[nB - kSwLeft + li];
    }
  }
  __syncthreads();
  DFTICK(3);
#undef DFTICK
  return !uni(sh.sw_fail);
}
