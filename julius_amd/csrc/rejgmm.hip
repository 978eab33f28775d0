// rejgmm.hip -- GMM-based input verification / rejection (-gmm, -gmmnum, -gmmreject) on gfx950.
//
// Replaces the scoring of libjulius/src/gmm.c: gmm_proceed() (gmm.c:574-600) scores every frame
// against a handful of one-state GMMs ("speech", "noise", ...) and adds the scores up per model;
// gmm_end() (gmm.c:614-660) then picks the winner.  gmm.c carries a private copy of the safe pruning
// that is NOT libsent's arithmetic: the Gaussians visited while the top-N list is not yet full are
// scored by gmm_compute_g_base() (gmm.c:177-194: squared distances summed from 0, gconst added
// last), the later ones by gmm_compute_g_safe() (gmm.c:218-240: gconst first; LOG_ZERO once the
// partial sum passes -2 x the list's last score -- the partial sums only grow, so that is decided
// by the final sum).  Everything else is cache_push() / addlog_array() as in calc_mix().
//
// The work is tiny (T x a few models x tens of Gaussians) and HBM-trivial; what matters is that it is
// the reference's number.  One thread per (frame, model); the running sums of an utterance are float
// additions in frame order (gmm.c:599), one thread per (utterance, model).
#include "jamd_device.h"

#include <cmath>
#include <string>
#include <vector>

struct jamd_rejgmm {
  jamd_engine *eng = nullptr;
  int D = 0, nmodel = 0, gprune_num = 0, maxmix = 0;
  float *d_mean = nullptr, *d_ivar = nullptr, *d_gconst = nullptr, *d_logw = nullptr;
  int *d_st_off = nullptr, *d_ent_dens = nullptr, *d_model_state = nullptr;
  int *d_utt_off = nullptr; int utt_cap = 0;
  std::vector<std::string> names;          // model names / is_voice: only known when loaded from a file
  std::vector<unsigned char> is_voice;
};

namespace {
using namespace jamd;

template <int NMAX>
__global__ void __launch_bounds__(256)
rejgmm_frame_kernel(const float *__restrict__ mean, const float *__restrict__ ivar, const float *__restrict__ gconst,
                    const int *__restrict__ st_off, const int *__restrict__ ent_dens, const float *__restrict__ logw,
                    const int *__restrict__ model_state, const float *__restrict__ frames,
                    const float *__restrict__ tbl, float *__restrict__ out, int T, int nmodel, int D,
                    int gprune_num, float addmin_f) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)T * nmodel) return;
  const int t = (int)(idx / nmodel), k = (int)(idx % nmodel);
  const float *__restrict__ vec = frames + (size_t)t * D;
  const int s = model_state[k];
  const int e0 = st_off[s], n = st_off[s + 1] - e0;
  const int cap = gprune_num < NMAX ? gprune_num : NMAX;           // list slots ever used (n <= NMAX)
  float sc[NMAX]; int id[NMAX];
  int len = 0;
#pragma unroll
  for (int i = 0; i < NMAX; i++) { sc[i] = JAMD_LOG_ZERO; id[i] = 0; }
  for (int i = 0; i < n; i++) {                                     // gmm_gprune_safe(), gmm.c:296-313
    const int g = ent_dens[e0 + i];
    float last = JAMD_LOG_ZERO;                                     // thres: the list's last score
#pragma unroll
    for (int j = 0; j < NMAX; j++) if (j == len - 1) last = sc[j];
    float score = JAMD_LOG_ZERO;
    if (len < gprune_num) {
      if (g >= 0) {                                                 // gmm_compute_g_base()
        const float *__restrict__ m = mean + (size_t)g * D, *__restrict__ v = ivar + (size_t)g * D;
        float tmp = 0.0f;
        for (int d = 0; d < D; d++) { const float x = vec[d] - m[d]; tmp += x * x * v[d]; }
        score = (tmp + gconst[g]) * -0.5f;
      }
    } else {
      if (g >= 0) {                                                 // gmm_compute_g_safe()
        const float *__restrict__ m = mean + (size_t)g * D, *__restrict__ v = ivar + (size_t)g * D;
        const float fthres = last * -2.0f;
        float tmp = gconst[g];
        for (int d = 0; d < D; d++) { const float x = vec[d] - m[d]; tmp += x * x * v[d]; }
        score = tmp > fthres ? JAMD_LOG_ZERO : tmp * -0.5f;
      }
      if (score <= last) continue;
    }
    topn_push<NMAX>(sc, id, len, cap, score, i);
  }
  // gmm_calc_mix(), gmm.c:335-370: weights of the survivors, log-sum from the last slot down
  float y = JAMD_LOG_ZERO;
#pragma unroll
  for (int i = NMAX - 1; i >= 0; i--) {
    if (i < len) {
      float w = 0.0f;
      // id[i] is a register value; the weight is a gather
      w = logw[e0 + id[i]];
      y = addlog_step(y, sc[i] + w, tbl, addmin_f);
    }
  }
  out[(size_t)t * nmodel + k] = finish_state(y);
}

// gmm_proceed(), gmm.c:599: gmm_score[k] += score, frame after frame
__global__ void __launch_bounds__(64)
rejgmm_sum_kernel(const float *__restrict__ fsc, const int *__restrict__ utt_off, float *__restrict__ out,
                  int nutt, int nmodel) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nutt * nmodel) return;
  const int u = idx / nmodel, k = idx % nmodel;
  float acc = 0.0f;
  for (int t = utt_off[u]; t < utt_off[u + 1]; t++) acc += fsc[(size_t)t * nmodel + k];
  out[idx] = acc;
}

template <typename T>
int upload(T **dst, const T *src, size_t n) {
  if (hipMalloc(dst, sizeof(T) * (n ? n : 1)) != hipSuccess) return JAMD_ENOMEM;
  if (n && hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice) != hipSuccess) return JAMD_ENODEV;
  return JAMD_OK;
}

}  // namespace

extern "C" {

int jamd_rejgmm_create(jamd_engine *e, const jamd_gmm_desc *gmm, const int *model_state, int nmodel, int gprune_num,
                       jamd_rejgmm **out) {
  if (!e || !gmm || !model_state || !out || nmodel < 1 || gprune_num < 1) { jamd_set_error("jamd_rejgmm_create: bad argument"); return JAMD_EINVAL; }
  *out = nullptr;
  if (gmm->nbook > 0 || gmm->nstream != 1) { jamd_set_error("jamd_rejgmm_create: tied-mixture and multi-stream GMMs are not supported (gmm_init(), gmm.c:431-434)"); return JAMD_EINVAL; }
  int maxmix = 1;
  for (int k = 0; k < nmodel; k++) {
    if (model_state[k] < 0 || model_state[k] >= gmm->nstate) { jamd_set_error("jamd_rejgmm_create: model_state[%d] out of range", k); return JAMD_EINVAL; }
    const int n = gmm->st_off[model_state[k] + 1] - gmm->st_off[model_state[k]];
    if (n > maxmix) maxmix = n;
  }
  const int need = gprune_num < maxmix ? gprune_num : maxmix;
  if (need > 64) { jamd_set_error("jamd_rejgmm_create: more than 64 Gaussians kept per model (-gmmnum %d, %d mixtures)", gprune_num, maxmix); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  jamd_rejgmm *m = new jamd_rejgmm();
  m->eng = e; m->D = gmm->veclen; m->nmodel = nmodel; m->gprune_num = gprune_num; m->maxmix = maxmix;
  int rc = upload(&m->d_mean, gmm->mean, (size_t)gmm->ndens * gmm->veclen);
  if (rc == JAMD_OK) rc = upload(&m->d_ivar, gmm->ivar, (size_t)gmm->ndens * gmm->veclen);
  if (rc == JAMD_OK) rc = upload(&m->d_gconst, gmm->gconst, (size_t)gmm->ndens);
  if (rc == JAMD_OK) rc = upload(&m->d_logw, gmm->ent_logw, (size_t)gmm->nentry);
  if (rc == JAMD_OK) rc = upload(&m->d_st_off, gmm->st_off, (size_t)gmm->nstate + 1);
  if (rc == JAMD_OK) rc = upload(&m->d_ent_dens, gmm->ent_dens, (size_t)gmm->nentry);
  if (rc == JAMD_OK) rc = upload(&m->d_model_state, model_state, (size_t)nmodel);
  if (rc != JAMD_OK) { jamd_set_error("jamd_rejgmm_create: device allocation or copy failed"); jamd_rejgmm_destroy(m); return rc; }
  *out = m;
  return JAMD_OK;
}

void jamd_rejgmm_destroy(jamd_rejgmm *m) {
  if (!m) return;
  (void)hipSetDevice(m->eng->device);
  void *ptrs[] = { m->d_mean, m->d_ivar, m->d_gconst, m->d_logw, m->d_st_off, m->d_ent_dens, m->d_model_state, m->d_utt_off };
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete m;
}

int jamd_rejgmm_nmodel(const jamd_rejgmm *m) { return m ? m->nmodel : 0; }

int jamd_rejgmm_set_models(jamd_rejgmm *m, const char *const *names, const unsigned char *is_voice) {
  if (!m || !names || !is_voice) { jamd_set_error("jamd_rejgmm_set_models: NULL argument"); return JAMD_EINVAL; }
  m->names.assign(names, names + m->nmodel);
  m->is_voice.assign(is_voice, is_voice + m->nmodel);
  return JAMD_OK;
}

const char *jamd_rejgmm_model_name(const jamd_rejgmm *m, int k) {
  return (m && k >= 0 && k < (int)m->names.size()) ? m->names[k].c_str() : "";
}

// gmm_end() (gmm.c:614-660) + gmm_valid_input() (:672-679) on one row of utterance sums
int jamd_rejgmm_verdict(const jamd_rejgmm *m, const float *utt_scores, int *winner, float *cm, int *accepted) {
  if (!m || !utt_scores || !winner || !cm || !accepted) { jamd_set_error("jamd_rejgmm_verdict: NULL argument"); return JAMD_EINVAL; }
  float maxprob = JAMD_LOG_ZERO; int maxid = 0; bool found = false;
  for (int i = 0; i < m->nmodel; i++) if (maxprob < utt_scores[i]) { maxprob = utt_scores[i]; maxid = i; found = true; }
  float sum = 0.0f;
  for (int i = 0; i < m->nmodel; i++) sum += (float)pow(10.0, 0.05 * (double)(utt_scores[i] - maxprob));
  *winner = maxid; *cm = (float)(1.0 / sum);
  *accepted = found && (m->is_voice.empty() || m->is_voice[maxid]) ? 1 : 0;
  return JAMD_OK;
}
int jamd_rejgmm_veclen(const jamd_rejgmm *m) { return m ? m->D : 0; }

int jamd_rejgmm_frame_scores_dev(jamd_rejgmm *m, const float *dev_frames, int T, float *dev_out, void *stream) {
  if (!m || !dev_frames || !dev_out || T < 0) { jamd_set_error("jamd_rejgmm_frame_scores_dev: bad argument"); return JAMD_EINVAL; }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(m->eng->device));
  hipStream_t st = jamd_stream(m->eng, stream);
  const long total = (long)T * m->nmodel;
  const dim3 grid((unsigned)((total + 255) / 256));
  const int need = m->gprune_num < m->maxmix ? m->gprune_num : m->maxmix;
#define JAMD_REJ(N)                                                                                      \
  hipLaunchKernelGGL((rejgmm_frame_kernel<N>), grid, dim3(256), 0, st, m->d_mean, m->d_ivar, m->d_gconst, \
                     m->d_st_off, m->d_ent_dens, m->d_logw, m->d_model_state, dev_frames, m->eng->d_addlog, \
                     dev_out, T, m->nmodel, m->D, m->gprune_num, m->eng->addmin_f)
  if (need <= 4) JAMD_REJ(4);
  else if (need <= 8) JAMD_REJ(8);
  else if (need <= 16) JAMD_REJ(16);
  else if (need <= 32) JAMD_REJ(32);
  else JAMD_REJ(64);
#undef JAMD_REJ
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { jamd_set_error("jamd_rejgmm_frame_scores_dev: launch failed: %s", hipGetErrorString(le)); return JAMD_ELAUNCH; }
  return JAMD_OK;
}

int jamd_rejgmm_utt_scores_dev(jamd_rejgmm *m, const float *dev_frame_scores, int T, const int *utt_off, int nutt,
                               float *dev_out, void *stream) {
  if (!m || !dev_frame_scores || !dev_out || !utt_off || nutt < 1 || T < 0) { jamd_set_error("jamd_rejgmm_utt_scores_dev: bad argument"); return JAMD_EINVAL; }
  if (utt_off[0] != 0 || utt_off[nutt] != T) { jamd_set_error("jamd_rejgmm_utt_scores_dev: utt_off must run from 0 to T"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(m->eng->device));
  hipStream_t st = jamd_stream(m->eng, stream);
  if (nutt + 1 > m->utt_cap) {
    if (m->d_utt_off) JAMD_HIP(hipFree(m->d_utt_off));
    m->d_utt_off = nullptr; m->utt_cap = 0;
    JAMD_HIP(hipMalloc(&m->d_utt_off, sizeof(int) * (nutt + 1)));
    m->utt_cap = nutt + 1;
  }
  JAMD_HIP(hipMemcpyAsync(m->d_utt_off, utt_off, sizeof(int) * (nutt + 1), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(rejgmm_sum_kernel, dim3((nutt * m->nmodel + 63) / 64), dim3(64), 0, st, dev_frame_scores,
                     m->d_utt_off, dev_out, nutt, m->nmodel);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { jamd_set_error("jamd_rejgmm_utt_scores_dev: launch failed: %s", hipGetErrorString(le)); return JAMD_ELAUNCH; }
  return JAMD_OK;
}

int jamd_rejgmm_scores_host(jamd_rejgmm *m, const float *host_frames, int T, const int *utt_off, int nutt,
                            float *host_frame_scores, float *host_utt_scores) {
  if (!m || !host_frames || T < 0 || (!host_frame_scores && !host_utt_scores) || (host_utt_scores && (!utt_off || nutt < 1))) {
    jamd_set_error("jamd_rejgmm_scores_host: bad argument"); return JAMD_EINVAL;
  }
  if (T == 0) {
    if (host_utt_scores) for (int i = 0; i < nutt * m->nmodel; i++) host_utt_scores[i] = 0.0f;
    return JAMD_OK;
  }
  JAMD_HIP(hipSetDevice(m->eng->device));
  hipStream_t st = m->eng->stream;
  float *d_fr = nullptr, *d_fs = nullptr, *d_us = nullptr;
  int rc = JAMD_OK;
  if (hipMalloc(&d_fr, sizeof(float) * (size_t)T * m->D) != hipSuccess ||
      hipMalloc(&d_fs, sizeof(float) * (size_t)T * m->nmodel) != hipSuccess ||
      (host_utt_scores && hipMalloc(&d_us, sizeof(float) * (size_t)nutt * m->nmodel) != hipSuccess)) {
    jamd_set_error("jamd_rejgmm_scores_host: out of device memory"); rc = JAMD_ENOMEM;
  }
  if (rc == JAMD_OK && hipMemcpyAsync(d_fr, host_frames, sizeof(float) * (size_t)T * m->D, hipMemcpyHostToDevice, st) != hipSuccess) {
    jamd_set_error("jamd_rejgmm_scores_host: copy failed"); rc = JAMD_ENODEV;
  }
  if (rc == JAMD_OK) rc = jamd_rejgmm_frame_scores_dev(m, d_fr, T, d_fs, st);
  if (rc == JAMD_OK && host_utt_scores) rc = jamd_rejgmm_utt_scores_dev(m, d_fs, T, utt_off, nutt, d_us, st);
  if (rc == JAMD_OK && host_frame_scores &&
      hipMemcpyAsync(host_frame_scores, d_fs, sizeof(float) * (size_t)T * m->nmodel, hipMemcpyDeviceToHost, st) != hipSuccess) rc = JAMD_ELAUNCH;
  if (rc == JAMD_OK && host_utt_scores &&
      hipMemcpyAsync(host_utt_scores, d_us, sizeof(float) * (size_t)nutt * m->nmodel, hipMemcpyDeviceToHost, st) != hipSuccess) rc = JAMD_ELAUNCH;
  if (rc == JAMD_OK && hipStreamSynchronize(st) != hipSuccess) { jamd_set_error("jamd_rejgmm_scores_host: device fault"); rc = JAMD_ELAUNCH; }
  if (d_fr) (void)hipFree(d_fr);
  if (d_fs) (void)hipFree(d_fs);
  if (d_us) (void)hipFree(d_us);
  return rc;
}

}  // extern "C"
