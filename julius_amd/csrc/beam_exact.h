// beam_exact.h -- interface between beam.hip (host side of the first pass) and beam_exact.hip (the
// exact-order frame-parallel kernel).  Internal, not installed.
#pragma once
#include "beam_common.h"

namespace jamdb {

struct XWork {
  Work w;                      // the slices of the frame-parallel kernel are reused as they are
  unsigned o_nodefirst;        // u32 [nnode]  ~(first visiting index) of a node whose cell overflowed to nodekey[]
  unsigned o_bitmap;           // u32 [gbm_words] creation-order bitmap when it outgrows LDS
  unsigned o_heap;             // u64 [tok_cap + 2] heap of a frame with more tokens than the LDS heap holds
  unsigned o_collect;          // u32x4 [beam + 256] wide layout: the top list on its way from the heap to the sorted lists
  unsigned o_sweep;            // xbeam_sweep_bytes(beam): scratch of the sweep replay (beam_sweep.h), 0 = none
  unsigned o_pstat;            // int [8]: how this utterance's pruning steps were resolved (jamd_beam_prune_stats())
  // multipath lexicons (beam_exact_mp.h)
  unsigned o_nodetok;          // u32 [n_mp_tgt] token id + 1 of the token a node holds, by the node's number among the nodes a root leads to (0 = none)
  unsigned o_arr;              // int [tok_cap] tindex[]: the frame's tokens as the mid-frame sort left them + the appended ones
  unsigned o_key2;             // u32 [tok_cap] their score bits in that arrangement (input of the frame's final cut)
  int mp;                      // 1 = multipath lexicon: beam_exact_mp_kernel
  unsigned o_mp_iso, o_mp_shared, o_mp_start;   // the roots' own transitions (int4 lists in the lexicon arena, jamd_lexicon)
  int n_mp_iso, n_mp_shared, n_mp_start;
  unsigned o_mp_tgt;           // int [nnode] in the lexicon arena: that number, -1 = no root leads to the node
  int n_mp_tgt;
  int nt, lds_budget;          // workgroup shape: threads, dynamic LDS it may use (full: NT / kMaxDynLds; half: kHalfNT / kHalfDynLds)
  int wide;                    // 1 = wide-beam layout: survivors in the utterance's slice (o_sv), the pruning step overlays
                               //     the whole LDS image but welist[] (see xbeam_layout())
  int cells_at, fixed_end;     // start of the per-launch part of the image (cells / pruning overlay / score row)
  int off_dov;                 // start of the pruning step's overlay (narrow: = cells_at; wide: behind welist[])
  int s1, xw;                  // visiting index = (source position << s1) | transition number; roots start at xw
  int nslot;                   // LDS Viterbi cells (16 bytes each: key, node, first visit)
  int bm_words;                // LDS bitmap capacity
  int heap_cap, b_cap;         // pruning step: LDS heap entries, top-k list entries
  int prune_mode;              // 0 = closed-form extraction, 1 = sequential extraction always (timing / test)
  // byte offsets in dynamic LDS
  int off_atom, off_we, off_dbase, off_tpre, off_bm, off_cells, off_lnode, off_lfirst, off_row;
  int off_compr, off_vpos, off_id, off_idt, off_hist, off_tail, off_heap;   // the pruning step's overlay
  int lds_bytes;
};

// Fills the LDS layout for beam width w.beam.  maxfan = 2 + most extra arcs of a node, nroot = startnum.
// 0 = ok, -1 = the visiting index does not fit 32 bits, -2 = the per-survivor arrays do not fit LDS (beam too wide
// even for the wide layout), -3 = more tokens per frame than the heap's position keys can number.
// half = the half shape: 512 threads and half a CU's LDS, so that two utterances share a CU and one's barriers and
// wave-serial sections overlap the other's work (-2 when a typical frame would not fit that image).
// mp = multipath lexicon (nroot = the roots a word end is followed by: isolated roots under an N-gram, all under a grammar).
int xbeam_layout(XWork *xw, const Work &w, int maxfan, int nroot, int ninit, int nshared, bool half, bool mp = false);
// places the per-launch part of the image (cells, pruning overlay, score row of nstate floats or none)
void xbeam_place(XWork *xw, int nstate);
hipError_t xbeam_prepare();
void xbeam_launch(const LexDev &lx, const XWork &xw, const float *scores, int nstate, const int *d_utt_off, int nutt,
                  int smode, bool timed, hipStream_t st);
void xbeam_prune_order_launch(const XWork &xw, const unsigned *d_keys, int n, int k, int *d_out, int *d_nout,
                              unsigned long long *d_hglob, u32x4 *d_collect, unsigned char *d_sweep, int *d_arr, hipStream_t st);
size_t xbeam_sweep_bytes(int beam);   // global scratch of the sweep replay per utterance

}  // namespace jamdb
