// gmm_outprob.hip -- GMM state-likelihood kernels for gfx950 (CDNA4).
//
// Replaces, for a whole block of frames at once, what the reference computes
// one (state, frame) at a time:
//   outprob_state() batch loop      libsent/src/phmm/outprob.c:230-242
//   calc_mix()                      libsent/src/phmm/calc_mix.c:41-81
//   gprune_none()/compute_g_base()  libsent/src/phmm/gprune_none.c:59-147
//   addlog_array()                  libsent/src/phmm/addlog.c:103-123
//
// Arithmetic contract (bit-exact with the reference's x86-64 object code):
//   * per Gaussian: tmp = gconst; for d ascending: x = o_d - mu_d;
//     tmp = tmp + (x*x)*ivar_d   -- four separately rounded fp32 operations,
//     this file is compiled with -ffp-contract=off so nothing is fused;
//     score = tmp * -0.5 ; score += ln w
//   * mixture log-sum: the table scan of addlog_array() from the LAST mixture
//     to the first, index (unsigned)((double)(-d) * 33333.3333 + 0.5)
//   * result (float)((double)lse * .434294482), LOG_ZERO if lse <= LOG_ZERO or
//     lse == 0 (calc_mix.c:78-80).
//
// Kernel "tile" (the batched hot kernel): ONE LANE = ONE FRAME (FPL frames per
// lane), so the feature vector lives in VGPRs, every Gaussian's (mean, ivar,
// gconst, ln w) record is wave-uniform and arrives through scalar loads into
// SGPRs (each VALU op takes it as its one SGPR operand), the D-loop is exactly
// 4 VALU ops per (frame, Gaussian, dim) with no LDS traffic, and the mixture
// log-sum is an in-lane scan in reference order -- no cross-lane reduction and
// no divergence except the (predicated) table gather.  Results are staged
// through a wave-private LDS tile so the [T][S] matrix is written in coalesced
// row segments.  See DESIGN.md "K1".
#include "jamd_internal.h"

namespace {

__device__ __forceinline__ float addlog_step(float y, float sc, const float *__restrict__ tbl,
                                             float addmin_f) {
  // addlog.c:108-121 with y the running value: larger stays in y.
  const bool gt = sc > y;
  const float hi = gt ? sc : y;
  const float lo = gt ? y : sc;
  const float d = lo - hi;
  float r = hi;
  if (!(d < addmin_f)) {
    const unsigned idx = (unsigned)((double)(-d) * JAMD_TMAG + 0.5);
    r = hi + tbl[idx];
  }
  return r;
}

__device__ __forceinline__ float finish_state(float lse) {
  // calc_mix.c:73-80, one stream, stream weight 1
  if (lse <= JAMD_LOG_ZERO || lse == 0.0f) return JAMD_LOG_ZERO;
  return (float)((double)lse * JAMD_INV_LOG_TEN);
}

constexpr int kWaves = 4;  // waves per workgroup

// XCD-aware block decode: the dispatcher places block b on XCD b % 8
// (MI355X_MICROARCH.md "Workgroup dispatch"); all frame-blocks of one state
// range are given the same b % 8 so the range's records stay in that XCD's L2.
__device__ __forceinline__ bool decode_block(int nfb, int nstb, int &fb, int &sb) {
  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  sb = xcd + 8 * (q / nfb);
  fb = q % nfb;
  return sb < nstb;
}

template <int D, int FPL, int NS>
__global__ void __launch_bounds__(64 * kWaves)
gmm_tile_kernel(const float *__restrict__ rec, const int *__restrict__ st_off,
                const float *__restrict__ frames, const float *__restrict__ tbl,
                float *__restrict__ out, int T, int S, int nsb, int nfb, int nstb,
                float addmin_f) {
  constexpr int REC = ((2 * D + 2) + 3) & ~3;
  constexpr int FPW = 64 * FPL;
  __shared__ float tile[kWaves][FPW][NS + 1];

  int fb, sb;
  if (!decode_block(nfb, nstb, fb, sb)) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int t0 = (fb * kWaves + wave) * FPW;
  if (t0 >= T) return;  // whole wave out of range (no block-level barriers below)

  float v[FPL][D];
#pragma unroll
  for (int k = 0; k < FPL; k++) {
    int t = t0 + k * 64 + lane;
    if (t > T - 1) t = T - 1;
    const float *fr = frames + (size_t)t * D;
#pragma unroll
    for (int d = 0; d < D; d++) v[k][d] = fr[d];
  }

  const int s_begin = sb * nsb;
  const int s_end = min(S, s_begin + nsb);
  for (int sg = s_begin; sg < s_end; sg += NS) {
    const int ns = min(NS, s_end - sg);
    for (int si = 0; si < ns; si++) {
      const int e0 = st_off[sg + si], e1 = st_off[sg + si + 1];
      float y[FPL];
#pragma unroll
      for (int k = 0; k < FPL; k++) y[k] = JAMD_LOG_ZERO;
      for (int e = e1 - 1; e >= e0; e--) {
        const float *__restrict__ r = rec + (size_t)e * REC;
        const float gc = r[2 * D], lw = r[2 * D + 1];
        float acc[FPL];
#pragma unroll
        for (int k = 0; k < FPL; k++) acc[k] = gc;
#pragma unroll
        for (int d = 0; d < D; d++) {
          const float mu = r[d], iv = r[D + d];
#pragma unroll
          for (int k = 0; k < FPL; k++) {
            float x = v[k][d] - mu;
            x = x * x;
            x = x * iv;
            acc[k] = acc[k] + x;
          }
        }
        const bool nulld = (gc != gc);  // NULL density marker (gprune_none.c:67)
#pragma unroll
        for (int k = 0; k < FPL; k++) {
          float sc = acc[k] * -0.5f;
          if (nulld) sc = JAMD_LOG_ZERO;
          sc = sc + lw;
          y[k] = addlog_step(y[k], sc, tbl, addmin_f);
        }
      }
#pragma unroll
      for (int k = 0; k < FPL; k++) tile[wave][k * 64 + lane][si] = finish_state(y[k]);
    }
    // wave-private tile: make the LDS writes visible to the other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int RPI = 64 / NS;  // rows per store instruction
    const int col = lane % NS, rsub = lane / NS;
#pragma unroll 4
    for (int it = 0; it < FPW / RPI; it++) {
      const int rr = it * RPI + rsub;
      const int t = t0 + rr;
      if (t < T && col < ns) out[(size_t)t * S + sg + col] = tile[wave][rr][col];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Generic-D variant: the frame vectors sit in LDS transposed [d][frame-in-wave]
// (conflict-free ds_read_b32), everything else as above.
template <int FPL, int NS>
__global__ void __launch_bounds__(64 * kWaves)
gmm_tile_generic_kernel(const float *__restrict__ rec, const int *__restrict__ st_off,
                        const float *__restrict__ frames, const float *__restrict__ tbl,
                        float *__restrict__ out, int T, int S, int D, int REC, int nsb, int nfb,
                        int nstb, float addmin_f) {
  constexpr int FPW = 64 * FPL;
  __shared__ float tile[kWaves][FPW][NS + 1];
  extern __shared__ __align__(16) float dyn[];  // [kWaves][D][FPW]

  int fb, sb;
  if (!decode_block(nfb, nstb, fb, sb)) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int t0 = (fb * kWaves + wave) * FPW;
  if (t0 >= T) return;
  float *vt = dyn + (size_t)wave * D * FPW;
  for (int k = 0; k < FPL; k++) {
    int t = t0 + k * 64 + lane;
    if (t > T - 1) t = T - 1;
    const float *fr = frames + (size_t)t * D;
    for (int d = 0; d < D; d++) vt[d * FPW + k * 64 + lane] = fr[d];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int s_begin = sb * nsb;
  const int s_end = min(S, s_begin + nsb);
  for (int sg = s_begin; sg < s_end; sg += NS) {
    const int ns = min(NS, s_end - sg);
    for (int si = 0; si < ns; si++) {
      const int e0 = st_off[sg + si], e1 = st_off[sg + si + 1];
      float y[FPL];
#pragma unroll
      for (int k = 0; k < FPL; k++) y[k] = JAMD_LOG_ZERO;
      for (int e = e1 - 1; e >= e0; e--) {
        const float *__restrict__ r = rec + (size_t)e * REC;
        const float gc = r[2 * D], lw = r[2 * D + 1];
        float acc[FPL];
#pragma unroll
        for (int k = 0; k < FPL; k++) acc[k] = gc;
        for (int d = 0; d < D; d++) {
          const float mu = r[d], iv = r[D + d];
#pragma unroll
          for (int k = 0; k < FPL; k++) {
            float x = vt[d * FPW + k * 64 + lane] - mu;
            x = x * x;
            x = x * iv;
            acc[k] = acc[k] + x;
          }
        }
        const bool nulld = (gc != gc);
#pragma unroll
        for (int k = 0; k < FPL; k++) {
          float sc = acc[k] * -0.5f;
          if (nulld) sc = JAMD_LOG_ZERO;
          sc = sc + lw;
          y[k] = addlog_step(y[k], sc, tbl, addmin_f);
        }
      }
#pragma unroll
      for (int k = 0; k < FPL; k++) tile[wave][k * 64 + lane][si] = finish_state(y[k]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int RPI = 64 / NS;
    const int col = lane % NS, rsub = lane / NS;
    for (int it = 0; it < FPW / RPI; it++) {
      const int rr = it * RPI + rsub;
      const int t = t0 + rr;
      if (t < T && col < ns) out[(size_t)t * S + sg + col] = tile[wave][rr][col];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int D, int FPL, int NS>
int launch_tile(jamd_gmm *g, const float *frames, int T, float *out, hipStream_t st) {
  constexpr int FPB = kWaves * 64 * FPL;
  const int nfb = (T + FPB - 1) / FPB;
  // states per block: aim for >= 8 blocks per CU, in multiples of NS
  int nsb = NS;
  const int want = g->eng->num_cu * 8;
  while (nsb < 16 * NS && (long)nfb * ((g->S + 2 * nsb - 1) / (2 * nsb)) >= want) nsb *= 2;
  const int nstb = (g->S + nsb - 1) / nsb;
  const int grid = 8 * ((nstb + 7) / 8) * nfb;
  hipLaunchKernelGGL((gmm_tile_kernel<D, FPL, NS>), dim3(grid), dim3(64 * kWaves), 0, st,
                     g->d_rec, g->d_st_off, frames, g->eng->d_addlog, out, T, g->S, nsb, nfb,
                     nstb, g->eng->addmin_f);
  snprintf(g->last_kernel, sizeof(g->last_kernel), "gmm_tile<D=%d,FPL=%d,NS=%d> grid=%d nsb=%d",
           D, FPL, NS, grid, nsb);
  return JAMD_OK;
}

template <int FPL, int NS>
int launch_tile_generic(jamd_gmm *g, const float *frames, int T, float *out, hipStream_t st) {
  constexpr int FPB = kWaves * 64 * FPL;
  const int nfb = (T + FPB - 1) / FPB;
  int nsb = NS;
  const int want = g->eng->num_cu * 8;
  while (nsb < 16 * NS && (long)nfb * ((g->S + 2 * nsb - 1) / (2 * nsb)) >= want) nsb *= 2;
  const int nstb = (g->S + nsb - 1) / nsb;
  const int grid = 8 * ((nstb + 7) / 8) * nfb;
  const size_t dyn = sizeof(float) * kWaves * g->D * 64 * FPL;
  hipLaunchKernelGGL((gmm_tile_generic_kernel<FPL, NS>), dim3(grid), dim3(64 * kWaves), dyn, st,
                     g->d_rec, g->d_st_off, frames, g->eng->d_addlog, out, T, g->S, g->D, g->rec,
                     nsb, nfb, nstb, g->eng->addmin_f);
  snprintf(g->last_kernel, sizeof(g->last_kernel), "gmm_tile_generic<FPL=%d,NS=%d> D=%d grid=%d",
           FPL, NS, g->D, grid);
  return JAMD_OK;
}

int ensure(float **p, size_t *cap, size_t need) {
  if (*cap >= need) return JAMD_OK;
  if (*p) JAMD_HIP(hipFree(*p));
  *p = nullptr; *cap = 0;
  JAMD_HIP(hipMalloc(p, need));
  *cap = need;
  return JAMD_OK;
}

}  // namespace

extern "C" {

int jamd_gmm_create(jamd_engine *e, const jamd_gmm_desc *d, int gprune, int gprune_num,
                    jamd_gmm **out) {
  if (!e || !d || !out) { jamd_set_error("jamd_gmm_create: NULL argument"); return JAMD_EINVAL; }
  *out = nullptr;
  if (d->nstream != 1) {
    jamd_set_error("jamd_gmm_create: nstream=%d; only single-stream models are supported", d->nstream);
    return JAMD_EINVAL;
  }
  if (d->nstate <= 0 || d->veclen <= 0 || d->veclen > 1024 || d->nentry < 0 || d->ndens < 0) {
    jamd_set_error("jamd_gmm_create: bad dimensions S=%d D=%d G=%d E=%d", d->nstate, d->veclen,
                   d->ndens, d->nentry);
    return JAMD_EINVAL;
  }
  if (gprune != JAMD_GPRUNE_NONE && gprune != JAMD_GPRUNE_SAFE) {
    jamd_set_error("jamd_gmm_create: gprune method %d is not implemented on the device "
                   "(heu/beam are frame-order dependent; use none or safe)", gprune);
    return JAMD_EINVAL;
  }
  if (!d->mean || !d->ivar || !d->gconst || !d->st_off || (d->nentry && (!d->ent_dens || !d->ent_logw))) {
    jamd_set_error("jamd_gmm_create: NULL model array");
    return JAMD_EINVAL;
  }
  if (d->st_off[0] != 0 || d->st_off[d->nstate] != d->nentry) {
    jamd_set_error("jamd_gmm_create: st_off must run from 0 to nentry");
    return JAMD_EINVAL;
  }
  bool any_tied = false;
  if (d->nbook > 0 && d->st_book)
    for (int s = 0; s < d->nstate; s++) any_tied |= (d->st_book[s] >= 0);
  if (any_tied) {
    jamd_set_error("jamd_gmm_create: tied-mixture states are not implemented yet");
    return JAMD_EINVAL;
  }
  if (gprune == JAMD_GPRUNE_SAFE) {
    jamd_set_error("jamd_gmm_create: gprune safe for plain states is not implemented yet");
    return JAMD_EINVAL;
  }
  JAMD_HIP(hipSetDevice(e->device));
  jamd_gmm *g = new jamd_gmm();
  g->eng = e; g->S = d->nstate; g->D = d->veclen; g->E = d->nentry; g->nbook = d->nbook;
  g->gprune = gprune; g->gprune_num = gprune_num;
  const int D = g->D;
  g->rec = ((2 * D + 2) + 3) & ~3;
  g->uniform_mix = true;
  for (int s = 0; s < g->S; s++) {
    const int n = d->st_off[s + 1] - d->st_off[s];
    if (n < 0) { delete g; jamd_set_error("jamd_gmm_create: st_off not monotone at %d", s); return JAMD_EINVAL; }
    if (n > g->maxmix) g->maxmix = n;
    if (n != d->st_off[1] - d->st_off[0]) g->uniform_mix = false;
  }
  // entry records: the state's densities laid out contiguously in state order so
  // the scalar stream of a state range is one linear read (shared ~m/~v macros
  // are duplicated -- 288 GB of HBM makes that free).
  std::vector<float> rec((size_t)g->E * g->rec, 0.0f);
  for (int en = 0; en < g->E; en++) {
    float *r = rec.data() + (size_t)en * g->rec;
    const int dn = d->ent_dens[en];
    if (dn >= d->ndens) { delete g; jamd_set_error("jamd_gmm_create: density index %d out of range", dn); return JAMD_EINVAL; }
    if (dn >= 0) {
      memcpy(r, d->mean + (size_t)dn * D, sizeof(float) * D);
      memcpy(r + D, d->ivar + (size_t)dn * D, sizeof(float) * D);
      r[2 * D] = d->gconst[dn];
    } else {
      r[2 * D] = __builtin_nanf("");
    }
    r[2 * D + 1] = d->ent_logw[en];
  }
  JAMD_HIP(hipMalloc(&g->d_rec, sizeof(float) * (rec.size() ? rec.size() : 4)));
  JAMD_HIP(hipMemcpy(g->d_rec, rec.data(), sizeof(float) * rec.size(), hipMemcpyHostToDevice));
  JAMD_HIP(hipMalloc(&g->d_st_off, sizeof(int) * (g->S + 1)));
  JAMD_HIP(hipMemcpy(g->d_st_off, d->st_off, sizeof(int) * (g->S + 1), hipMemcpyHostToDevice));
  *out = g;
  return JAMD_OK;
}

void jamd_gmm_destroy(jamd_gmm *g) {
  if (!g) return;
  (void)hipSetDevice(g->eng->device);
  void *ptrs[] = { g->d_rec, g->d_st_off, g->d_st_book, g->d_book_off, g->d_book_rec,
                   g->d_ent_logw, g->d_frames, g->d_out, g->d_tm_score, g->d_tm_id, g->d_tm_num };
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete g;
}

int jamd_gmm_nstate(const jamd_gmm *g) { return g ? g->S : -1; }
int jamd_gmm_veclen(const jamd_gmm *g) { return g ? g->D : -1; }
const char *jamd_gmm_last_kernel(const jamd_gmm *g) { return g ? g->last_kernel : ""; }

int jamd_gmm_outprob_dev(jamd_gmm *g, const float *dev_frames, int T, float *dev_out, void *stream) {
  if (!g || !dev_frames || !dev_out || T < 0) {
    jamd_set_error("jamd_gmm_outprob_dev: bad argument");
    return JAMD_EINVAL;
  }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(g->eng->device));
  hipStream_t st = jamd_stream(g->eng, stream);
  int rc;
  switch (g->D) {
    case 39: rc = launch_tile<39, 2, 16>(g, dev_frames, T, dev_out, st); break;
    case 38: rc = launch_tile<38, 2, 16>(g, dev_frames, T, dev_out, st); break;
    case 26: rc = launch_tile<26, 2, 16>(g, dev_frames, T, dev_out, st); break;
    case 25: rc = launch_tile<25, 2, 16>(g, dev_frames, T, dev_out, st); break;
    default: rc = launch_tile_generic<2, 16>(g, dev_frames, T, dev_out, st); break;
  }
  if (rc != JAMD_OK) return rc;
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    jamd_set_error("jamd_gmm_outprob_dev: launch failed: %s", hipGetErrorString(le));
    return JAMD_ELAUNCH;
  }
  return JAMD_OK;
}

int jamd_gmm_outprob_host(jamd_gmm *g, const float *host_frames, int T, float *host_out) {
  if (!g || !host_frames || !host_out || T < 0) {
    jamd_set_error("jamd_gmm_outprob_host: bad argument");
    return JAMD_EINVAL;
  }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(g->eng->device));
  int rc;
  if ((rc = ensure(&g->d_frames, &g->frames_cap, sizeof(float) * (size_t)T * g->D)) != JAMD_OK) return rc;
  if ((rc = ensure(&g->d_out, &g->out_cap, sizeof(float) * (size_t)T * g->S)) != JAMD_OK) return rc;
  hipStream_t st = g->eng->stream;
  JAMD_HIP(hipMemcpyAsync(g->d_frames, host_frames, sizeof(float) * (size_t)T * g->D,
                          hipMemcpyHostToDevice, st));
  if ((rc = jamd_gmm_outprob_dev(g, g->d_frames, T, g->d_out, st)) != JAMD_OK) return rc;
  JAMD_HIP(hipMemcpyAsync(host_out, g->d_out, sizeof(float) * (size_t)T * g->S,
                          hipMemcpyDeviceToHost, st));
  hipError_t se = hipStreamSynchronize(st);
  if (se != hipSuccess) {
    jamd_set_error("jamd_gmm_outprob_host: execution failed: %s", hipGetErrorString(se));
    return JAMD_ELAUNCH;
  }
  return JAMD_OK;
}

int jamd_gmm_tmix_cache_dev(jamd_gmm *, const float *, int, float *, int *, int *, void *) {
  jamd_set_error("jamd_gmm_tmix_cache_dev: tied-mixture path not implemented yet");
  return JAMD_EINVAL;
}

}  // extern "C"
