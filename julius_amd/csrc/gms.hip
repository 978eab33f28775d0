// gms.hip -- Gaussian mixture selection (-gshmm / -gsnum) on gfx950.
//
// Replaces gms_state() and its helpers (libsent/src/phmm/gms.c:189-412, gms_gprune.c:80-257):
// with a selection model loaded, Julius scores a small monophone GMM set ("GS HMM") every frame,
// keeps the nbest highest states and, for every state of the real model whose selection state is
// NOT among them, returns the selection state's score instead of the real one.  On a CPU that saves
// most of the Gaussian work; here every real score exists anyway and this stage REPLACES the ones
// Julius would not have computed, so that a configuration with -gshmm gives the reference's
// numbers.  Two sequential dependencies are kept exactly:
//   * compute_g_max() evaluates last frame's best Gaussian of the state first and then the others
//     from the highest index down with a strict >, so an exact tie is broken by history;
//   * the nbest states are taken from a partial heap sort over an index array that is not reset
//     between frames (sort_gsindex_upward()), so a tie on the selection boundary is, too.
// Hence one wave per utterance walks its frames in order: the lanes share the states of a frame
// (per-Gaussian scores come from gmm_dens, the kernel behind the plugin slot; rows reach LDS by DMA
// one frame ahead).  The selection itself has two forms, like the first pass: by default every lane
// ranks its states against all others -- the reference's set unless two selection states tie exactly
// on the boundary -- and in strict order (jamd_gms_set_strict_order) lane 0 runs the reference's
// heap in LDS.  Even then the index array is reset at every utterance (the reference never resets
// it), which again can only matter for an exact tie on the boundary.
#include "jamd_device.h"

struct jamd_gms {
  jamd_engine *eng = nullptr;
  jamd_gmm *gs = nullptr;          // the selection model (per-Gaussian scores)
  int Sgs = 0, Egs = 0, S = 0, nbest = 0, strict = 0;
  int *d_st_off = nullptr;         // [Sgs + 1]
  float *d_logw = nullptr;         // [Egs]
  int *d_state2gs = nullptr;       // [S]
  int *d_utt_off = nullptr; int utt_cap = 0;
  float *d_dens = nullptr; size_t dens_cap = 0;
  float *d_fs = nullptr; size_t fs_cap = 0;
};

namespace {
using namespace jamd;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave per utterance.  LDS: fs[Sgs] | idx[Sgs] | last[Sgs] | st_off[Sgs+1] | logw[Egs] | row[2][EgsPad].
// The frame's per-Gaussian scores (one row of dens) arrive by LDS-DMA, the next frame's row while
// this one is processed, so the serial walk over the frames never waits for HBM.
// STRICT: lane 0 runs the reference's heap.  Otherwise every lane ranks its own states against all
// (broadcast reads), which selects the same set unless two states tie exactly on the boundary; then
// the lower state id wins where the reference's answer depends on the heap's history.
// VAR 1 (the form that is launched; VAR 0 is the scalar ranking loop it replaced, 9.2 vs 7.3 ms): the two latency chains of the default
// form -- one LDS read per ranking step, one per Gaussian in the max -- are batched four wide.
template <bool STRICT, int VAR>
__global__ void __launch_bounds__(64)
gms_select_kernel(const float *__restrict__ dens, const int *__restrict__ g_st_off, const float *__restrict__ g_logw,
                  const int *__restrict__ utt_off, float *__restrict__ fs_out, int Sgs, int Egs, int EgsPad, int nbest) {
  extern __shared__ float lds[];
  float *fs = lds;                              // [Sgs]
  int *idx = (int *)(fs + Sgs);                 // [Sgs]   heap order (STRICT) / selected flag
  int *last = idx + Sgs;                        // [Sgs]
  int *st_off = last + Sgs;                     // [Sgs + 1]
  float *logw = (float *)(st_off + Sgs + 1);    // [Egs]
  float *rowbuf = logw + Egs;                   // [2][EgsPad], 256-byte aligned by the launcher's padding
  const int lane = threadIdx.x, u = blockIdx.x;
  const int t0 = utt_off[u], t1 = utt_off[u + 1];
  rowbuf += (64 - ((3 * Sgs + Sgs + 1 + Egs) & 63)) & 63;
  for (int i = lane; i < Sgs; i += 64) { idx[i] = i; last[i] = -1; }
  for (int i = lane; i <= Sgs; i += 64) st_off[i] = g_st_off[i];
  for (int i = lane; i < Egs; i += 64) logw[i] = g_logw[i];
  auto stage = [&](int buf, int t) {
    const float *g = dens + (size_t)t * Egs + lane;
    float *dst = rowbuf + buf * EgsPad;
    for (int e0 = 0; e0 < EgsPad; e0 += 64)
      __builtin_amdgcn_global_load_lds((glb_void *)(g + e0), (lds_void *)(dst + e0), 4, 0, 0);
  };
  if (t0 < t1) stage(0, t0);
  int cur = 0;
  for (int t = t0; t < t1; t++, cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this frame's row has landed
    wave_sync();
    if (t + 1 < t1) stage(cur ^ 1, t + 1);
    const float *row = rowbuf + cur * EgsPad;
    for (int i = lane; i < Sgs; i += 64) {                         // compute_g_max(), LAST_BEST
      const int e0 = st_off[i], n = st_off[i + 1] - e0;
      const int first = (last[i] != -1) ? last[i] : n - 1;
      // calc_contprob_with_safe_pruning(): a score below the running maximum (LOG_ZERO for the first
      // one) comes back as LOG_ZERO
      float maxprob = row[e0 + first];
      if (maxprob < JAMD_LOG_ZERO) maxprob = JAMD_LOG_ZERO;
      int maxi = first;
      if constexpr (VAR == 1) {
        int k = n - 1;
        for (; k >= 3; k -= 4) {                                     // same visiting order, four loads in flight
          const float p0 = row[e0 + k], p1 = row[e0 + k - 1], p2 = row[e0 + k - 2], p3 = row[e0 + k - 3];
          if (k != first && p0 > maxprob) { maxprob = p0; maxi = k; }
          if (k - 1 != first && p1 > maxprob) { maxprob = p1; maxi = k - 1; }
          if (k - 2 != first && p2 > maxprob) { maxprob = p2; maxi = k - 2; }
          if (k - 3 != first && p3 > maxprob) { maxprob = p3; maxi = k - 3; }
        }
        for (; k >= 0; k--) {
          const float p = row[e0 + k];
          if (k != first && p > maxprob) { maxprob = p; maxi = k; }
        }
      } else {
        for (int k = n - 1; k >= 0; k--) {
          const float p = row[e0 + k];
          if (k != first && p > maxprob) { maxprob = p; maxi = k; }
        }
      }
      last[i] = maxi;
      float sum = 0.0f;
      sum += (maxprob + logw[e0 + maxi]) * 1.0f;
      fs[i] = (float)((double)sum * JAMD_INV_LOG_TEN);
    }
    wave_sync();
    if (STRICT) {
      if (lane == 0) {                                              // sort_gsindex_upward() + do_gms()
        const int totalnum = Sgs, neednum = nbest < Sgs ? nbest : Sgs;
#define SD_(A) idx[(A) - 1]
#define SV_(A) (fs[idx[(A) - 1]])
        for (int root = totalnum / 2; root >= 1; root--) {
          const int sd = SD_(root);
          int parent = root, child;
          while ((child = parent * 2) <= totalnum) {
            if (child < totalnum && SV_(child) < SV_(child + 1)) child++;
            if (fs[sd] >= SV_(child)) break;
            SD_(parent) = SD_(child); parent = child;
          }
          SD_(parent) = sd;
        }
        int n = totalnum;
        while (n > totalnum - neednum) {
          const int sd = SD_(n);
          SD_(n) = SD_(1); n--;
          int parent = 1, child;
          while ((child = parent * 2) <= n) {
            if (child < n && SV_(child) < SV_(child + 1)) child++;
            if (fs[sd] >= SV_(child)) break;
            SD_(parent) = SD_(child); parent = child;
          }
          SD_(parent) = sd;
        }
#undef SD_
#undef SV_
        for (int i = totalnum - neednum; i < totalnum; i++) fs[idx[i]] = JAMD_LOG_ZERO;   // selected
      }
      wave_sync();
      for (int i = lane; i < Sgs; i += 64) fs_out[(size_t)t * Sgs + i] = fs[i];
    } else {
      for (int i = lane; i < Sgs; i += 64) {                        // rank of state i among all
        const float v = fs[i];
        int rank = 0;
        if constexpr (VAR == 1) {
          const float4 *f4 = reinterpret_cast<const float4 *>(fs);       // fs sits at LDS offset 0
          int j = 0;
          for (; j + 4 <= Sgs; j += 4) {
            const float4 w = f4[j >> 2];
            rank += (w.x > v || (w.x == v && j < i)) ? 1 : 0;
            rank += (w.y > v || (w.y == v && j + 1 < i)) ? 1 : 0;
            rank += (w.z > v || (w.z == v && j + 2 < i)) ? 1 : 0;
            rank += (w.w > v || (w.w == v && j + 3 < i)) ? 1 : 0;
          }
          for (; j < Sgs; j++) {
            const float w = fs[j];
            rank += (w > v || (w == v && j < i)) ? 1 : 0;
          }
        } else {
          for (int j = 0; j < Sgs; j++) {
            const float w = fs[j];
            rank += (w > v || (w == v && j < i)) ? 1 : 0;
          }
        }
        fs_out[(size_t)t * Sgs + i] = rank < nbest ? JAMD_LOG_ZERO : v;
      }
    }
  }
}

// gms_state(): fallback value unless the selection state was selected (marked LOG_ZERO)
__global__ void __launch_bounds__(256)
gms_combine_kernel(const float *__restrict__ fs, const int *__restrict__ state2gs, float *__restrict__ scores,
                   int T, int S, int Sgs) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int g = state2gs[s];
  if (g < 0) return;
  for (int t = blockIdx.y; t < T; t += gridDim.y) {
    const float f = fs[(size_t)t * Sgs + g];
    if (f != JAMD_LOG_ZERO) scores[(size_t)t * S + s] = f;
  }
}

int grow(float **p, size_t *cap, size_t need) {
  if (*cap >= need) return JAMD_OK;
  if (*p) JAMD_HIP(hipFree(*p));
  *p = nullptr; *cap = 0;
  JAMD_HIP(hipMalloc(p, need));
  *cap = need;
  return JAMD_OK;
}

}  // namespace

extern "C" {

int jamd_gms_create(jamd_engine *e, const jamd_gmm_desc *gs, const int *state2gs, int nstate, int nbest,
                    jamd_gms **out) {
  if (!e || !gs || !state2gs || !out || nstate <= 0 || nbest < 1) { jamd_set_error("jamd_gms_create: bad argument"); return JAMD_EINVAL; }
  *out = nullptr;
  if (gs->nbook > 0 || gs->nstream != 1) { jamd_set_error("jamd_gms_create: the selection model must be a plain single-stream GMM"); return JAMD_EINVAL; }
  for (int s = 0; s < nstate; s++)
    if (state2gs[s] >= gs->nstate) { jamd_set_error("jamd_gms_create: state2gs[%d] out of range", s); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  jamd_gms *m = new jamd_gms();
  m->eng = e; m->Sgs = gs->nstate; m->Egs = gs->nentry; m->S = nstate; m->nbest = nbest;
  int rc = jamd_gmm_create(e, gs, JAMD_GPRUNE_NONE, 0, &m->gs);
  if (rc == JAMD_OK && hipMalloc(&m->d_st_off, sizeof(int) * (gs->nstate + 1)) != hipSuccess) rc = JAMD_ENOMEM;
  if (rc == JAMD_OK && hipMalloc(&m->d_logw, sizeof(float) * (gs->nentry > 0 ? gs->nentry : 1)) != hipSuccess) rc = JAMD_ENOMEM;
  if (rc == JAMD_OK && hipMalloc(&m->d_state2gs, sizeof(int) * nstate) != hipSuccess) rc = JAMD_ENOMEM;
  if (rc == JAMD_OK &&
      (hipMemcpy(m->d_st_off, gs->st_off, sizeof(int) * (gs->nstate + 1), hipMemcpyHostToDevice) != hipSuccess ||
       hipMemcpy(m->d_logw, gs->ent_logw, sizeof(float) * gs->nentry, hipMemcpyHostToDevice) != hipSuccess ||
       hipMemcpy(m->d_state2gs, state2gs, sizeof(int) * nstate, hipMemcpyHostToDevice) != hipSuccess)) rc = JAMD_ENODEV;
  if (rc != JAMD_OK) { if (rc == JAMD_ENOMEM) jamd_set_error("jamd_gms_create: out of device memory"); jamd_gms_destroy(m); return rc; }
  *out = m;
  return JAMD_OK;
}

int jamd_gms_nstate(const jamd_gms *m) { return m ? m->S : 0; }

int jamd_gms_set_strict_order(jamd_gms *m, int on) {
  if (!m) { jamd_set_error("jamd_gms_set_strict_order: NULL"); return JAMD_EINVAL; }
  m->strict = on != 0;
  return JAMD_OK;
}

void jamd_gms_destroy(jamd_gms *m) {
  if (!m) return;
  (void)hipSetDevice(m->eng->device);
  if (m->gs) jamd_gmm_destroy(m->gs);
  void *ptrs[] = { m->d_st_off, m->d_logw, m->d_state2gs, m->d_utt_off, m->d_dens, m->d_fs };
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete m;
}

int jamd_gms_apply_dev(jamd_gms *m, const float *dev_frames, int T, const int *utt_off, int nutt,
                       float *dev_scores, void *stream) {
  if (!m || !dev_frames || !dev_scores || T < 0 || (utt_off && nutt < 1)) { jamd_set_error("jamd_gms_apply_dev: bad argument"); return JAMD_EINVAL; }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(m->eng->device));
  hipStream_t st = jamd_stream(m->eng, stream);
  const int one[2] = {0, T};
  if (!utt_off) { utt_off = one; nutt = 1; }
  if (utt_off[0] != 0 || utt_off[nutt] != T) { jamd_set_error("jamd_gms_apply_dev: utt_off must run from 0 to T"); return JAMD_EINVAL; }
  int rc;
  if (nutt + 1 > m->utt_cap) {
    if (m->d_utt_off) JAMD_HIP(hipFree(m->d_utt_off));
    m->d_utt_off = nullptr;
    JAMD_HIP(hipMalloc(&m->d_utt_off, sizeof(int) * (nutt + 1)));
    m->utt_cap = nutt + 1;
  }
  if ((rc = grow(&m->d_dens, &m->dens_cap, sizeof(float) * ((size_t)T * m->Egs + 64))) != JAMD_OK) return rc;
  if ((rc = grow(&m->d_fs, &m->fs_cap, sizeof(float) * (size_t)T * m->Sgs)) != JAMD_OK) return rc;
  JAMD_HIP(hipMemcpyAsync(m->d_utt_off, utt_off, sizeof(int) * (nutt + 1), hipMemcpyHostToDevice, st));
  if ((rc = jamd_gmm_dens_dev(m->gs, dev_frames, T, m->d_dens, st)) != JAMD_OK) return rc;
  const int EgsPad = (m->Egs + 63) & ~63;
  const size_t lds = sizeof(float) * ((size_t)3 * m->Sgs + m->Sgs + 1 + m->Egs + 64 + 2 * (size_t)EgsPad);
  if (lds > 159 * 1024) { jamd_set_error("jamd_gms_apply_dev: a selection model of %d states / %d Gaussians does not fit in LDS", m->Sgs, m->Egs); return JAMD_EINVAL; }
  // the four-wide ranking loop (template argument 1) measured 7.3 ms against 9.2 ms for the scalar one
  // (profiles/r02a_gms_timing_variants.json) and is the only form launched
  auto kern = m->strict ? gms_select_kernel<true, 1> : gms_select_kernel<false, 1>;
  if (lds > 48 * 1024) JAMD_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(nutt), dim3(64), lds, st, m->d_dens, m->d_st_off, m->d_logw, m->d_utt_off,
                     m->d_fs, m->Sgs, m->Egs, EgsPad, m->nbest);
  hipLaunchKernelGGL(gms_combine_kernel, dim3((m->S + 255) / 256, T < 1024 ? T : 1024), dim3(256), 0, st, m->d_fs,
                     m->d_state2gs, dev_scores, T, m->S, m->Sgs);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { jamd_set_error("jamd_gms_apply_dev: launch failed: %s", hipGetErrorString(le)); return JAMD_ELAUNCH; }
  return JAMD_OK;
}

int jamd_gms_apply_host(jamd_gms *m, const float *host_frames, int T, const int *utt_off, int nutt,
                        float *host_scores) {
  if (!m || !host_frames || !host_scores || T < 0) { jamd_set_error("jamd_gms_apply_host: bad argument"); return JAMD_EINVAL; }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(m->eng->device));
  hipStream_t st = m->eng->stream;
  const int D = jamd_gmm_veclen(m->gs);
  float *d_fr = nullptr, *d_sc = nullptr;
  int rc = JAMD_OK;
  if (hipMalloc(&d_fr, sizeof(float) * (size_t)T * D) != hipSuccess ||
      hipMalloc(&d_sc, sizeof(float) * (size_t)T * m->S) != hipSuccess) {
    jamd_set_error("jamd_gms_apply_host: out of device memory"); rc = JAMD_ENOMEM;
  }
  if (rc == JAMD_OK && (hipMemcpyAsync(d_fr, host_frames, sizeof(float) * (size_t)T * D, hipMemcpyHostToDevice, st) != hipSuccess ||
                        hipMemcpyAsync(d_sc, host_scores, sizeof(float) * (size_t)T * m->S, hipMemcpyHostToDevice, st) != hipSuccess)) {
    jamd_set_error("jamd_gms_apply_host: copy failed"); rc = JAMD_ENODEV;
  }
  if (rc == JAMD_OK) rc = jamd_gms_apply_dev(m, d_fr, T, utt_off, nutt, d_sc, st);
  if (rc == JAMD_OK && (hipMemcpyAsync(host_scores, d_sc, sizeof(float) * (size_t)T * m->S, hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess)) { jamd_set_error("jamd_gms_apply_host: copy failed"); rc = JAMD_ELAUNCH; }
  if (d_fr) (void)hipFree(d_fr);
  if (d_sc) (void)hipFree(d_sc);
  return rc;
}

}  // extern "C"
