// jamd_internal.h -- shared internals of the gfx950 engine (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "julius_amd.h"

// addlog.c:28-30
#define JAMD_TBLSIZE 500000
#define JAMD_TMAG 33333.3333
// stddefs.h:176 / :111 (double constants in the reference)
#define JAMD_LOG_ADDMIN (-13.815510558)
#define JAMD_INV_LOG_TEN (.434294482)
// calc_dnn.c:344-347
#define JAMD_LOGISTIC_FACTOR 20000
#define JAMD_LOGISTIC_MAX (16 * JAMD_LOGISTIC_FACTOR)

void jamd_set_error(const char *fmt, ...);

#define JAMD_HIP(call)                                                            \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      jamd_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),       \
                     __FILE__, __LINE__);                                         \
      return (e_ == hipErrorOutOfMemory) ? JAMD_ENOMEM : JAMD_ENODEV;             \
    }                                                                             \
  } while (0)

struct jamd_engine {
  int device = 0;
  hipStream_t stream = nullptr;   // engine-owned stream
  float *d_addlog = nullptr;      // [JAMD_TBLSIZE + 1]; the extra last entry is 0.0f
  float *d_logistic = nullptr;    // [JAMD_LOGISTIC_MAX + 1]
  float addmin_f = 0.f;           // smallest float >= LOG_ADDMIN (exact float form of the double compare)
  int num_cu = 256;
};

struct jamd_gmm {
  jamd_engine *eng = nullptr;
  int S = 0, D = 0, E = 0, nbook = 0;
  int gprune = 0, gprune_num = 0;
  int rec = 0;                    // floats per entry record
  int maxmix = 0;
  bool uniform_mix = false;       // every state has the same entry count
  // device model
  float *d_rec = nullptr;         // [E][rec]: mean[D], ivar[D], gconst, logw
  int *d_st_off = nullptr;        // [S+1] original entry offsets (index d_ent_logw)
  int *d_st_off_plain = nullptr;  // [S+1] offsets into d_rec; a tied-mixture state has an empty range
  int E_plain = 0;
  // tied-mixture
  int *d_st_book = nullptr;       // [S]
  int *d_book_off = nullptr;      // [nbook+1] into book records
  std::vector<int> h_book_off;    // host copy of the same
  float *d_book_rec = nullptr;    // [sum book sizes][rec] (logw unused)
  float *d_ent_logw = nullptr;    // [E] entry weights (tied states index by codebook position)
  int *d_tied_states = nullptr;   // [ntied] ids of tied-mixture states
  int ntied = 0;
  int maxbook = 0;                // largest codebook
  int tm_cap = 0;                 // slots per (frame, book) in the codebook cache
  bool has_null = false;          // some mixture entry names no density (NULL density): K1 keeps its LOG_ZERO selects
  int hist_method = 0;            // JAMD_GPRUNE_HEU / _BEAM over tied-mixture codebooks (history pruning), else 0
  int *d_cur_utt_off = nullptr;   // [cur_nutt + 1] utterance boundaries of the running call (history pruning restarts
  int cur_nutt = 0; size_t utt_off_cap = 0;   //   at every utterance's first frame)
  // scratch
  float *d_frames = nullptr; size_t frames_cap = 0;
  float *d_out = nullptr; size_t out_cap = 0;
  float *d_tm_score = nullptr; int *d_tm_id = nullptr; int *d_tm_num = nullptr;
  float *d_narrow = nullptr; size_t narrow_cap = 0;   // [kNarrowT][E_plain] weighted Gaussian scores of a narrow call (K1n, gmm_outprob.hip)
  size_t tm_cap_bytes = 0, tm_id_bytes = 0, tm_num_bytes = 0;
  char last_kernel[64] = {0};
  // pinned staging copy of the running call's utterance boundaries (history pruning only) and the event behind its
  // upload: the buffer is rewritten only when the copy that read it is done
  int *h_utt_off = nullptr; size_t h_utt_off_cap = 0; hipEvent_t ev_utt_off = nullptr;
};

int jamd_gmm_launch_safe(jamd_gmm *g, const float *frames, int T, float *out, hipStream_t st);
int jamd_gmm_launch_tmix(jamd_gmm *g, const float *frames, int T, float *out, float *c_score,
                         int *c_id, int *c_num, hipStream_t st);

// A kernel whose static + dynamic LDS passes the default 64 KB window needs the attribute raised, and the sum
// must fit the 160 KB of a CU (the generic-D kernels keep 2 KB of frame data per vector component in LDS).
static inline int jamd_reserve_dyn_lds(const void *kernel, size_t dyn, const char *what) {
  if (dyn == 0) return JAMD_OK;
  hipFuncAttributes fa;
  JAMD_HIP(hipFuncGetAttributes(&fa, kernel));
  if (fa.sharedSizeBytes + dyn > 160u * 1024u) {
    jamd_set_error("%s: the vector length needs %zu bytes of LDS per workgroup (%zu static + %zu), the CU has 163840",
                   what, fa.sharedSizeBytes + dyn, (size_t)fa.sharedSizeBytes, dyn);
    return JAMD_EINVAL;
  }
  JAMD_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  return JAMD_OK;
}

// The first-pass tables of a binary N-gram file (mkbingram v5), as jamd_lexicon_desc carries them (csrc/readers.hip)
#include <string>
#include <vector>
struct JamdNgramTables {
  int mode = 0, nword = 0, nbigram = 0, n = 0, dir = 0;
  std::vector<float> uni_prob, uni_bo, bi_prob;
  std::vector<int> bi_bgn, bi_num, bi_wid;
  std::string names;                     // the vocabulary, every name followed by NUL, in N-gram id order
};
bool jamd_read_bingram_tables(const char *path, JamdNgramTables &out);

static inline hipStream_t jamd_stream(jamd_engine *e, void *s) {
  return s ? (hipStream_t)s : e->stream;
}
