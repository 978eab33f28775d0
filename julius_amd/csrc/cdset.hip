// cdset.hip -- pseudo-phone state-set scores (K5).
//
// Replaces outprob_cd() (libsent/src/phmm/outprob.c:383-400) and its three
// reductions over a CD_State_Set (htk_hmm.h:249-253), reading one row of the
// [T][S] state-score matrix instead of calling outprob_state() per member:
//   IWCD_MAX    outprob_cd_max()    outprob.c:332-344   running maximum from LOG_ZERO
//   IWCD_AVG    outprob_cd_avg()    outprob.c:356-370   float sum of members > LOG_ZERO,
//                                                        in member order, / (float)count
//   IWCD_NBEST  outprob_cd_nbest()  outprob.c:287-321   descending insertion list of at
//                                                        most N, summed best-first, / (float)n
// An empty / all-LOG_ZERO set yields 0/0 = NaN for avg and nbest exactly as the
// reference's float division does.
// One thread per (frame, set); consecutive threads take consecutive sets so
// the [T][nset] output is written coalesced; member gathers hit L2.
#include "jamd_device.h"

struct jamd_cdset {
  jamd_engine *eng = nullptr;
  int nset = 0, nstates = 0, method = 0, nbest = 0, maxset = 0;
  int *d_off = nullptr, *d_states = nullptr;
};

namespace {

using jamd::kNbestMax;

__global__ void __launch_bounds__(256)
cdset_kernel(const float *__restrict__ scores, const int *__restrict__ off,
             const int *__restrict__ states, float *__restrict__ cd, int T, int S, int nset,
             int method, int nbest) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nset) return;
  for (int t = blockIdx.y; t < T; t += gridDim.y) {         // gridDim.y is capped (65535 limit): frames are strided
    const float *row = scores + (size_t)t * S;
    const float r = jamd::cd_reduce(row, states, off[i], off[i + 1], method, nbest);
    cd[(size_t)t * nset + i] = r;
  }
}

}  // namespace

extern "C" {

int jamd_cdset_create(jamd_engine *e, int nset, const int *set_off, const int *states, int method,
                      int nbest, jamd_cdset **out) {
  if (!e || !out || nset < 0 || (nset && (!set_off || !states))) {
    jamd_set_error("jamd_cdset_create: bad argument");
    return JAMD_EINVAL;
  }
  *out = nullptr;
  if (method != JAMD_IWCD_MAX && method != JAMD_IWCD_AVG && method != JAMD_IWCD_NBEST) {
    jamd_set_error("jamd_cdset_create: unknown method %d", method);
    return JAMD_EINVAL;
  }
  if (method == JAMD_IWCD_NBEST && (nbest < 1 || nbest > kNbestMax)) {
    jamd_set_error("jamd_cdset_create: nbest=%d outside [1,%d]", nbest, kNbestMax);
    return JAMD_EINVAL;
  }
  if (nset && set_off[0] != 0) { jamd_set_error("jamd_cdset_create: set_off[0] != 0"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  jamd_cdset *c = new jamd_cdset();
  c->eng = e; c->nset = nset; c->method = method; c->nbest = nbest;
  c->nstates = nset ? set_off[nset] : 0;
  JAMD_HIP(hipMalloc(&c->d_off, sizeof(int) * (nset + 1)));
  JAMD_HIP(hipMalloc(&c->d_states, sizeof(int) * (c->nstates ? c->nstates : 1)));
  if (nset) {
    JAMD_HIP(hipMemcpy(c->d_off, set_off, sizeof(int) * (nset + 1), hipMemcpyHostToDevice));
    JAMD_HIP(hipMemcpy(c->d_states, states, sizeof(int) * c->nstates, hipMemcpyHostToDevice));
  }
  *out = c;
  return JAMD_OK;
}

void jamd_cdset_destroy(jamd_cdset *c) {
  if (!c) return;
  (void)hipSetDevice(c->eng->device);
  if (c->d_off) (void)hipFree(c->d_off);
  if (c->d_states) (void)hipFree(c->d_states);
  delete c;
}

int jamd_cdset_nset(const jamd_cdset *c) { return c ? c->nset : -1; }

int jamd_cdset_outprob_dev(jamd_cdset *c, const float *dev_scores, int T, int nstate, float *dev_cd,
                           void *stream) {
  if (!c || !dev_scores || !dev_cd || T < 0 || nstate <= 0) {
    jamd_set_error("jamd_cdset_outprob_dev: bad argument");
    return JAMD_EINVAL;
  }
  if (T == 0 || c->nset == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(c->eng->device));
  const dim3 grid((c->nset + 255) / 256, T < 4096 ? T : 4096);
  hipLaunchKernelGGL(cdset_kernel, grid, dim3(256), 0, jamd_stream(c->eng, stream), dev_scores,
                     c->d_off, c->d_states, dev_cd, T, nstate, c->nset, c->method, c->nbest);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    jamd_set_error("jamd_cdset_outprob_dev: launch failed: %s", hipGetErrorString(le));
    return JAMD_ELAUNCH;
  }
  return JAMD_OK;
}

}  // extern "C"
