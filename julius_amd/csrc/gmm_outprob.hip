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
#include "jamd_device.h"

namespace {
using namespace jamd;

#ifndef JAMD_GMM_LOGSUM_R4
#define JAMD_GMM_LOGSUM_R4 0
#endif
constexpr int kWaves = 4;  // waves per workgroup
int ensure(float **p, size_t *cap, size_t need);

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

// FPL (frames per lane) is even: frames are held as packed pairs so that the
// D-loop compiles to v_pk_add_f32 / v_pk_mul_f32 with the Gaussian's scalar
// broadcast through op_sel from an SGPR pair.  Measured on MI355X
// (tools/ubench_valu.hip): a plain VOP2 with an SGPR operand issues at ~0.6x
// the VGPR-only rate, the packed forms do not pay that penalty.
// The mixture log-sum is software-pipelined: the table gather for entry e is
// issued after its D-loop and consumed after the D-loop of entry e-1, so its
// latency hides behind ~160 packed VALU ops.
// HAS_NULL: the model holds NULL densities (gconst stored as NaN, gprune_none.c:67) -- only then does the log-sum step
// carry the two selects that turn such a score into LOG_ZERO (jamd_gmm::has_null, set when the records are packed).
template <int D, int FPL, int NS, bool HAS_NULL>
__global__ void __launch_bounds__(64 * kWaves)
gmm_tile_kernel(const float *__restrict__ rec, const int *__restrict__ st_off,
                const float *__restrict__ frames, const float *__restrict__ tbl,
                float *__restrict__ out, int T, int S, int nsb, int nfb, int nstb,
                float addmin_f) {
  static_assert(FPL % 2 == 0, "frames are processed as packed pairs");
  constexpr int REC = ((2 * D + 2) + 3) & ~3;
  constexpr int FPW = 64 * FPL;
  constexpr int NP = FPL / 2;
  __shared__ float tile[kWaves][FPW][NS + 1];

  int fb, sb;
  if (!decode_block(nfb, nstb, fb, sb)) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int t0 = (fb * kWaves + wave) * FPW;
  if (t0 >= T) return;  // whole wave out of range (no block-level barriers below)

  f2 v[NP][D];
#pragma unroll
  for (int p = 0; p < NP; p++) {
    int ta = t0 + (2 * p) * 64 + lane, tb = ta + 64;
    if (ta > T - 1) ta = T - 1;
    if (tb > T - 1) tb = T - 1;
    const float *fa = frames + (size_t)ta * D, *fb_ = frames + (size_t)tb * D;
#pragma unroll
    for (int d = 0; d < D; d++) { v[p][d].x = fa[d]; v[p][d].y = fb_[d]; }
  }

  // addlog table as a raw buffer (gfx9 dword 3: 32-bit data format); offsets past its JAMD_TBLSIZE + 1 entries read 0.
  // NaN inputs: a NaN |s - y| selects slot TBLSIZE (0.0f) and v_max_f32 drops a NaN operand -- scores are finite or
  // LOG_ZERO on this path (NULL densities are marked in gconst and handled before the log-sum), so no NaN reaches it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "gmm_tile: the raw buffer descriptor below is the gfx9 (CDNA) format; this library is written for gfx950 only"
#endif
  const __amdgpu_buffer_rsrc_t tbl_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tbl, 0, 4 * (JAMD_TBLSIZE + 1), 0x00020000);
  const int s_begin = sb * nsb;
  const int s_end = min(S, s_begin + nsb);
  for (int sg = s_begin; sg < s_end; sg += NS) {
    const int ns = min(NS, s_end - sg);
    for (int si = 0; si < ns; si++) {
      const int e0 = st_off[sg + si], e1 = st_off[sg + si + 1];
      f2 y2[NP], tv2[NP];            // running log-sum and pending table term, packed per frame pair
#pragma unroll
      for (int p = 0; p < NP; p++) { y2[p] = f2{JAMD_LOG_ZERO, JAMD_LOG_ZERO}; tv2[p] = f2{0.0f, 0.0f}; }
      for (int e = e1 - 1; e >= e0; e--) {
        const float *__restrict__ r = rec + (size_t)e * REC;
        const float gc = r[2 * D], lw = r[2 * D + 1];
        f2 acc[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) acc[p] = f2{gc, gc};
#pragma unroll
        for (int d = 0; d < D; d++) {
          const float mu = r[d], iv = r[D + d];
          const f2 mu2 = {mu, mu}, iv2 = {iv, iv};
#pragma unroll
          for (int p = 0; p < NP; p++) {
            f2 x = v[p][d] - mu2;
            x = x * x;
            x = x * iv2;
            acc[p] = acc[p] + x;
          }
        }
        const bool nulld = HAS_NULL && (gc != gc);  // NULL density marker (gprune_none.c:67)
        const float naddmin = -addmin_f;
#pragma unroll
        for (int p = 0; p < NP; p++) {
          // Packed form of addlog_step() for the two frames of a lane.  The table term of the
          // previous entry arrives in tv2 (0.0f from the extra table entry when none was due), so
          // finishing that step (addlog.c:119: y += tbl[idx]) is one packed add without a select.
          // The larger / smaller term of addlog.c:110-116 in the form with the fewest instructions (round 5: 29 -> 22
          // per Gaussian and frame pair): hi = max(s, y) (when they are equal either is the same float), and
          // -(lo - hi) = |s - y| exactly (a float subtraction commutes up to the sign), so the difference is ONE packed
          // subtraction whose absolute value enters the f32 -> f64 conversion as a source modifier; "d < LOG_ADDMIN"
          // becomes |s - y| > -LOG_ADDMIN on the same rounded values.  Index arithmetic in double as the reference.
#if JAMD_GMM_LOGSUM_R4       /* development A/B: the round-4 form of the step (29 instructions) */
          f2 s2 = acc[p] * f2{-0.5f, -0.5f};
          if (nulld) s2 = f2{JAMD_LOG_ZERO, JAMD_LOG_ZERO};
          s2 = s2 + f2{lw, lw};
          __builtin_amdgcn_sched_barrier(0);
          const f2 yy = y2[p] + tv2[p];
          const bool g0 = s2.x > yy.x, g1 = s2.y > yy.y;
          const f2 hi = {g0 ? s2.x : yy.x, g1 ? s2.y : yy.y};
          const f2 lo = {g0 ? yy.x : s2.x, g1 ? yy.y : s2.y};
          const f2 dd = lo - hi;
          const unsigned i0 = !(dd.x < addmin_f) ? (unsigned)((double)(-dd.x) * JAMD_TMAG + 0.5) : (unsigned)JAMD_TBLSIZE;
          const unsigned i1 = !(dd.y < addmin_f) ? (unsigned)((double)(-dd.y) * JAMD_TMAG + 0.5) : (unsigned)JAMD_TBLSIZE;
          tv2[p] = f2{tbl[i0], tbl[i1]};
          (void)naddmin; (void)tbl_rsrc;
#else
          f2 s2 = acc[p] * f2{-0.5f, -0.5f};
          if (nulld) s2 = f2{JAMD_LOG_ZERO, JAMD_LOG_ZERO};
          s2 = s2 + f2{lw, lw};
          __builtin_amdgcn_sched_barrier(0);     // keep the wait for the gathered term behind the D-loop
          const f2 yy = y2[p] + tv2[p];
          const f2 hi = {__builtin_fmaxf(s2.x, yy.x), __builtin_fmaxf(s2.y, yy.y)};
          const f2 df = s2 - yy;
          const float a0 = __builtin_fabsf(df.x), a1 = __builtin_fabsf(df.y);
          // the gather is a raw buffer load: table descriptor in four SGPRs, a 32-bit byte offset per lane, no 64-bit
          // address arithmetic (slot JAMD_TBLSIZE of the table holds 0.0f: "no table term")
          unsigned o0 = ((unsigned)((double)a0 * JAMD_TMAG + 0.5)) << 2, o1 = ((unsigned)((double)a1 * JAMD_TMAG + 0.5)) << 2;
          o0 = (a0 <= naddmin) ? o0 : 4u * (unsigned)JAMD_TBLSIZE;
          o1 = (a1 <= naddmin) ? o1 : 4u * (unsigned)JAMD_TBLSIZE;
          tv2[p] = f2{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(tbl_rsrc, (int)o0, 0, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(tbl_rsrc, (int)o1, 0, 0))};
#endif
          y2[p] = hi;
        }
      }
#pragma unroll
      for (int k = 0; k < FPL; k++) {
        const f2 fin = y2[k / 2] + tv2[k / 2];
        tile[wave][k * 64 + lane][si] = finish_state((k & 1) ? fin.y : fin.x);
      }
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
  // States per block: NS, the width of the output tile -- the smallest there is.  A block's fixed cost (its 128
  // frames per wave loaded into registers) is nothing against 16 states x 16 Gaussians of arithmetic, and short
  // blocks are what keeps the last wave of blocks of a launch short: at 64 000 frames a launch is four rounds of
  // 96-state blocks or twenty-three rounds of 16-state ones, 10.97 vs 9.93 ms (sweep on MI355X: 16 / 32 / 48 / 64 /
  // 96 / 128 states -> 9.93 / 10.02 / 10.35 / 10.62 / 10.97 / 11.22 ms; 384 000 frames: 60.4 vs 61.8 ms; 16 000:
  // 2.63 vs 2.78 ms).  Block b runs on XCD b % 8 and a state range is pinned to one XCD (decode_block), so the
  // blocks that share a range share an L2.
  const int nsb = NS;
  const int nstb = (g->S + nsb - 1) / nsb;
  const int grid = 8 * ((nstb + 7) / 8) * nfb;
  if (g->has_null)
    hipLaunchKernelGGL((gmm_tile_kernel<D, FPL, NS, true>), dim3(grid), dim3(64 * kWaves), 0, st,
                       g->d_rec, g->d_st_off_plain, frames, g->eng->d_addlog, out, T, g->S, nsb, nfb,
                       nstb, g->eng->addmin_f);
  else
    hipLaunchKernelGGL((gmm_tile_kernel<D, FPL, NS, false>), dim3(grid), dim3(64 * kWaves), 0, st,
                       g->d_rec, g->d_st_off_plain, frames, g->eng->d_addlog, out, T, g->S, nsb, nfb,
                       nstb, g->eng->addmin_f);
  snprintf(g->last_kernel, sizeof(g->last_kernel), "gmm_tile<D=%d,FPL=%d,NS=%d> grid=%d nsb=%d", D, FPL, NS, grid, nsb);
  return JAMD_OK;
}

template <int FPL, int NS>
int launch_tile_generic(jamd_gmm *g, const float *frames, int T, float *out, hipStream_t st) {
  constexpr int FPB = kWaves * 64 * FPL;
  const int nfb = (T + FPB - 1) / FPB;
  const int nsb = NS;   // as launch_tile()
  const int nstb = (g->S + nsb - 1) / nsb;
  const int grid = 8 * ((nstb + 7) / 8) * nfb;
  const size_t dyn = sizeof(float) * kWaves * g->D * 64 * FPL;
  const int rc = jamd_reserve_dyn_lds((const void *)gmm_tile_generic_kernel<FPL, NS>, dyn, "GMM outprob");
  if (rc != JAMD_OK) return rc;
  hipLaunchKernelGGL((gmm_tile_generic_kernel<FPL, NS>), dim3(grid), dim3(64 * kWaves), dyn, st,
                     g->d_rec, g->d_st_off_plain, frames, g->eng->d_addlog, out, T, g->S, g->D, g->rec,
                     nsb, nfb, nstb, g->eng->addmin_f);
  snprintf(g->last_kernel, sizeof(g->last_kernel), "gmm_tile_generic<FPL=%d,NS=%d> D=%d grid=%d",
           FPL, NS, g->D, grid);
  return JAMD_OK;
}

// ---------------------------------------------------------------------------------------
// K1n, the NARROW form of K1 (round 6): a call of a handful of frames -- live input, a streaming chunk
// (JAMD_STREAM_CHUNK = 25), the calcmix slot of a frame-synchronous caller.  K1 maps one lane to one frame (128 frame
// slots per wave): a 25-frame call fills a fifth of ONE wave per 16-state block, and every block still walks its 256
// Gaussians one after the other -- 250 us per call whatever T is, the 15.4 MB model streamed at 60 GB/s.  Here the
// mapping is turned round: one lane = one MIXTURE ENTRY (its record in 2 D + 2 registers, read once per call), the
// frames are wave-uniform and arrive by scalar load, two at a time as the packed pair of the D-loop; the weighted
// Gaussian scores go to a [T][E] scratch (coalesced), and a second kernel -- one lane per (frame, state) -- runs the
// table log-sum from the last mixture to the first (addlog_array(), addlog.c:103) and calc_mix()'s tail.  Same four
// separately rounded fp32 operations per dimension, same scan order: bit-identical to K1 (tests/test_gmm_gpu.py).
constexpr int kNarrowT = 256;      // calls of at most this many frames (the scratch is kNarrowT x E floats)
constexpr int kNarrowFB = 8;       // frames per block of the first kernel
template <int D, bool HAS_NULL>
__global__ void __launch_bounds__(64)
gmm_narrow_dens_kernel(const float *__restrict__ rec, const float *__restrict__ frames, float *__restrict__ dens, int T, int E,
                       int nfb, int neb) {
  constexpr int REC = (2 * D + 2 + 3) & ~3;
  // the frame blocks of one range of 64 entries run on ONE XCD (block b is placed on XCD b % 8): the range's records
  // come from HBM / Infinity Cache once and from that XCD's L2 for the other frame blocks (decode_block(), as K1)
  int fbk, ebk;
  if (!decode_block(nfb, neb, fbk, ebk)) return;
  const int e = ebk * 64 + threadIdx.x;
  const float4 *__restrict__ r4 = reinterpret_cast<const float4 *>(rec + (size_t)(e < E ? e : E - 1) * REC);
  float r[REC];
#pragma unroll
  for (int q = 0; q < REC / 4; q++) { const float4 v = r4[q]; r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w; }
  const float gc = r[2 * D], lw = r[2 * D + 1];
  const bool nulld = HAS_NULL && (gc != gc);      // NULL density marker (gprune_none.c:67)
  // a block takes kNarrowFB frames: with one wave per SIMD the scalar loads of a frame pair (a microsecond) were not
  // covered by anything (25 frames: 26 us); three or four waves per SIMD cover them (the record is read once per block: L2)
  const int t_end = min(T, (fbk + 1) * kNarrowFB);
  for (int t = fbk * kNarrowFB; t < t_end; t += 2) {
    const float *__restrict__ fa = frames + (size_t)t * D;
    const float *__restrict__ fb = frames + (size_t)(t + 1 < T ? t + 1 : t) * D;
    f2 acc = {gc, gc};
#pragma unroll
    for (int d = 0; d < D; d++) {
      f2 x = f2{fa[d], fb[d]} - f2{r[d], r[d]};
      x = x * x;
      x = x * f2{r[D + d], r[D + d]};
      acc = acc + x;
    }
    f2 s2 = acc * f2{-0.5f, -0.5f};
    if (nulld) s2 = f2{JAMD_LOG_ZERO, JAMD_LOG_ZERO};
    s2 = s2 + f2{lw, lw};
    if (e < E) {
      dens[(size_t)t * E + e] = s2.x;
      if (t + 1 < T) dens[(size_t)(t + 1) * E + e] = s2.y;
    }
  }
}

__global__ void __launch_bounds__(256)
gmm_narrow_lse_kernel(const float *__restrict__ dens, const int *__restrict__ st_off, const float *__restrict__ tbl,
                      float *__restrict__ out, int T, int S, int E, float addmin_f) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T * S) return;
  const int t = i / S, s = i - t * S;
  const int e0 = st_off[s], e1 = st_off[s + 1];
  const float *__restrict__ dr = dens + (size_t)t * E;
  float y = JAMD_LOG_ZERO;
  int e = e1 - 1;
  for (; e - 7 >= e0; e -= 8) {                    // eight terms in flight in front of the serial table scan
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = dr[e - q];
#pragma unroll
    for (int q = 0; q < 8; q++) y = addlog_step(y, v[q], tbl, addmin_f);
  }
  for (; e >= e0; e--) y = addlog_step(y, dr[e], tbl, addmin_f);
  out[(size_t)t * S + s] = finish_state(y);
}

// JAMD_GMM_NARROW=0: every call through K1 (A/B switch; read once)
bool narrow_enabled() {
  static const bool on = [] { const char *v = getenv("JAMD_GMM_NARROW"); return !(v && v[0] == '0'); }();
  return on;
}

template <int D>
int launch_narrow(jamd_gmm *g, const float *frames, int T, float *out, hipStream_t st) {
  const int E = g->E_plain;
  int rc = ensure(&g->d_narrow, &g->narrow_cap, sizeof(float) * (size_t)kNarrowT * (size_t)E);   // (first narrow call only)
  if (rc != JAMD_OK) return rc;
  const int grid = (E + 63) / 64;
  const int nfb = (T + kNarrowFB - 1) / kNarrowFB;
  const dim3 gr(8 * ((grid + 7) / 8) * nfb);
  if (g->has_null) hipLaunchKernelGGL((gmm_narrow_dens_kernel<D, true>), gr, dim3(64), 0, st, g->d_rec, frames, g->d_narrow, T, E, nfb, grid);
  else hipLaunchKernelGGL((gmm_narrow_dens_kernel<D, false>), gr, dim3(64), 0, st, g->d_rec, frames, g->d_narrow, T, E, nfb, grid);
  hipLaunchKernelGGL(gmm_narrow_lse_kernel, dim3((T * g->S + 255) / 256), dim3(256), 0, st, g->d_narrow, g->d_st_off_plain,
                     g->eng->d_addlog, out, T, g->S, E, g->eng->addmin_f);
  snprintf(g->last_kernel, sizeof(g->last_kernel), "gmm_narrow<D=%d> grid=%d + lse", D, grid);
  return JAMD_OK;
}

// ---------------------------------------------------------------------------------------
// Per-Gaussian scores for the reference's plugin slot (compute_gaussset / calcmix,
// plugin/calcmix.c:86-323): dens[t][e] = compute_g_base() of mixture entry e at frame t,
// (gconst + sum_d (o_d - mu_d)^2 * ivar_d) * -0.5, the same four fp32 operations per dimension as
// K1 and no weight, no log-sum (calc_mix() applies those to what the plugin returns).  One wave =
// 64 frames (their vectors transposed in LDS so any D fits), a block walks a chunk of entries,
// 16 at a time through a wave-private tile so that [T][E] is written in 64-byte row segments.
constexpr int kDensChunk = 1024;
__global__ void __launch_bounds__(64)
gmm_dens_kernel(const float *__restrict__ rec, const float *__restrict__ frames, float *__restrict__ out,
                int T, int E, int D, int REC) {
  extern __shared__ float xs[];                  // [D][64] then tile [64][17]
  float (*tile)[17] = reinterpret_cast<float (*)[17]>(xs + (size_t)D * 64);
  const int lane = threadIdx.x;
  const int t0 = blockIdx.x * 64, e_begin = blockIdx.y * kDensChunk;
  const int e_end = min(E, e_begin + kDensChunk);
  {
    int t = t0 + lane; if (t > T - 1) t = T - 1;
    const float *f = frames + (size_t)t * D;
    for (int d = 0; d < D; d++) xs[d * 64 + lane] = f[d];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int e0 = e_begin; e0 < e_end; e0 += 16) {
    const int ne = min(16, e_end - e0);
    for (int g = 0; g < ne; g++) {
      const float *__restrict__ r = rec + (size_t)(e0 + g) * REC;
      const float gc = r[2 * D];
      float acc = gc;
      for (int d = 0; d < D; d++) {
        float x = xs[d * 64 + lane] - r[d];
        x = x * x;
        x = x * r[D + d];
        acc = acc + x;
      }
      float sc = acc * -0.5f;
      if (gc != gc) sc = JAMD_LOG_ZERO;            // NULL density (gprune_none.c:67, plugin/calcmix.c:104)
      tile[lane][g] = sc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int col = lane & 15, rsub = lane >> 4;
    for (int it = 0; it < 16; it++) {
      const int rr = it * 4 + rsub, t = t0 + rr;
      if (t < T && col < ne) out[(size_t)t * E + e0 + col] = tile[rr][col];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
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

void jamd_gmm_destroy(jamd_gmm *g);
// the allocations of jamd_gmm_create(); on any failure the caller releases *gp through jamd_gmm_destroy()
static int gmm_create_impl(jamd_engine *e, const jamd_gmm_desc *d, int gprune, int gprune_num, jamd_gmm **gp) {
  jamd_gmm **out = gp;
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
  if (gprune != JAMD_GPRUNE_NONE && gprune != JAMD_GPRUNE_SAFE && gprune != JAMD_GPRUNE_HEU && gprune != JAMD_GPRUNE_BEAM) {
    jamd_set_error("jamd_gmm_create: unknown gprune method %d", gprune);
    return JAMD_EINVAL;
  }
  const bool history_pruning = gprune == JAMD_GPRUNE_HEU || gprune == JAMD_GPRUNE_BEAM;
  const int requested_gprune = gprune;
  // heu / beam on plain mixture states: calc_mix() passes last_id == NULL, the branch that is safe pruning
  // (gprune_heu.c:337-350, gprune_beam.c:337-350) -- same kernel, same numbers.  Checked below once the
  // states are classified.
  if (history_pruning) gprune = JAMD_GPRUNE_SAFE;
  if (!d->mean || !d->ivar || !d->gconst || !d->st_off || (d->nentry && (!d->ent_dens || !d->ent_logw))) {
    jamd_set_error("jamd_gmm_create: NULL model array");
    return JAMD_EINVAL;
  }
  if (d->st_off[0] != 0 || d->st_off[d->nstate] != d->nentry) {
    jamd_set_error("jamd_gmm_create: st_off must run from 0 to nentry");
    return JAMD_EINVAL;
  }
  JAMD_HIP(hipSetDevice(e->device));
  jamd_gmm *g = new jamd_gmm();
  *gp = g;                             // owned by the caller from here on
  g->eng = e; g->S = d->nstate; g->D = d->veclen; g->E = d->nentry;
  g->gprune = gprune; g->gprune_num = gprune_num;
  const int D = g->D;
  g->rec = ((2 * D + 2) + 3) & ~3;
  const bool have_books = d->nbook > 0 && d->st_book;
  std::vector<int> st_off_plain(g->S + 1, 0), tied;
  std::vector<int> book_first(d->nbook > 0 ? d->nbook : 0, -1);
  for (int s = 0; s < g->S; s++) {
    const int n = d->st_off[s + 1] - d->st_off[s];
    if (n < 0) { jamd_set_error("jamd_gmm_create: st_off not monotone at %d", s); return JAMD_EINVAL; }
    const int b = have_books ? d->st_book[s] : -1;
    if (b >= d->nbook) { jamd_set_error("jamd_gmm_create: codebook id %d out of range", b); return JAMD_EINVAL; }
    if (b >= 0) {
      tied.push_back(s);
      if (book_first[b] < 0) book_first[b] = s;
      else if (n != d->st_off[book_first[b] + 1] - d->st_off[book_first[b]]) {
        jamd_set_error("jamd_gmm_create: states of codebook %d disagree on its size", b); return JAMD_EINVAL;
      }
      st_off_plain[s + 1] = st_off_plain[s];
    } else {
      if (n > g->maxmix) g->maxmix = n;
      st_off_plain[s + 1] = st_off_plain[s] + n;
    }
  }
  g->E_plain = st_off_plain[g->S];
  g->ntied = (int)tied.size();
  // heu / beam over tied-mixture codebooks: frame t's thresholds come from the codebook's cached winners of frame
  // t - 1 (calc_tied_mix.c:203-215).  The device scores every state of every frame, so that history is the previous
  // frame of the same utterance: parity is defined against the reference under eager scoring
  // (outprob_set_batch_computation, outprob.c:230-242); see tmix_book_hist_kernel.
  if (history_pruning && g->ntied > 0) g->hist_method = requested_gprune;
  g->nbook = g->ntied ? d->nbook : 0;
  if (gprune == JAMD_GPRUNE_SAFE && gprune_num < 1) {
    jamd_set_error("jamd_gmm_create: gprune safe needs gprune_num >= 1"); return JAMD_EINVAL;
  }
  if (gprune == JAMD_GPRUNE_SAFE && gprune_num > 64) {
    jamd_set_error("jamd_gmm_create: gprune_num %d > 64 is not supported on the device", gprune_num);
    return JAMD_EINVAL;
  }
  auto fill_rec = [&](float *r, int dn, float lw) -> bool {
    if (dn >= d->ndens) return false;
    if (dn >= 0) {
      memcpy(r, d->mean + (size_t)dn * D, sizeof(float) * D);
      memcpy(r + D, d->ivar + (size_t)dn * D, sizeof(float) * D);
      r[2 * D] = d->gconst[dn];
      if (r[2 * D] != r[2 * D]) g->has_null = true;   // (a NaN gconst keeps the meaning it always had here)
    } else {
      r[2 * D] = __builtin_nanf("");   // NULL density (gprune_none.c:67)
      g->has_null = true;
    }
    r[2 * D + 1] = lw;
    return true;
  };
  // entry records of the plain states, contiguous in state order so the scalar
  // stream of a state range is one linear read (shared ~m/~v macros are
  // duplicated -- 288 GB of HBM makes that free).
  std::vector<float> rec((size_t)g->E_plain * g->rec, 0.0f);
  for (int s = 0; s < g->S; s++) {
    if (have_books && d->st_book[s] >= 0) continue;
    for (int k = 0; k < d->st_off[s + 1] - d->st_off[s]; k++) {
      const int en = d->st_off[s] + k;
      if (!fill_rec(rec.data() + (size_t)(st_off_plain[s] + k) * g->rec, d->ent_dens[en], d->ent_logw[en])) {
        jamd_set_error("jamd_gmm_create: density index %d out of range", d->ent_dens[en]); return JAMD_EINVAL;
      }
    }
  }
  JAMD_HIP(hipMalloc(&g->d_rec, sizeof(float) * (rec.size() ? rec.size() : 4)));
  JAMD_HIP(hipMemcpy(g->d_rec, rec.data(), sizeof(float) * rec.size(), hipMemcpyHostToDevice));
  JAMD_HIP(hipMalloc(&g->d_st_off, sizeof(int) * (g->S + 1)));
  JAMD_HIP(hipMemcpy(g->d_st_off, d->st_off, sizeof(int) * (g->S + 1), hipMemcpyHostToDevice));
  JAMD_HIP(hipMalloc(&g->d_st_off_plain, sizeof(int) * (g->S + 1)));
  JAMD_HIP(hipMemcpy(g->d_st_off_plain, st_off_plain.data(), sizeof(int) * (g->S + 1), hipMemcpyHostToDevice));
  if (g->ntied) {
    // codebooks: the densities of book b in codebook order are the entries of any
    // state tied to it (GCODEBOOK.d[], htk_hmm.h:196-201)
    std::vector<int> book_off(g->nbook + 1, 0);
    for (int b = 0; b < g->nbook; b++) {
      const int n = book_first[b] >= 0 ? d->st_off[book_first[b] + 1] - d->st_off[book_first[b]] : 0;
      book_off[b + 1] = book_off[b] + n;
      if (n > g->maxbook) g->maxbook = n;
    }
    std::vector<float> brec((size_t)book_off[g->nbook] * g->rec, 0.0f);
    for (int b = 0; b < g->nbook; b++) {
      if (book_first[b] < 0) continue;
      for (int k = 0; k < book_off[b + 1] - book_off[b]; k++) {
        if (!fill_rec(brec.data() + (size_t)(book_off[b] + k) * g->rec,
                      d->ent_dens[d->st_off[book_first[b]] + k], 0.0f)) {
          jamd_set_error("jamd_gmm_create: codebook density index out of range"); return JAMD_EINVAL;
        }
      }
    }
    g->tm_cap = (gprune == JAMD_GPRUNE_NONE) ? g->maxbook : (gprune_num < g->maxbook ? gprune_num : g->maxbook);
    JAMD_HIP(hipMalloc(&g->d_book_rec, sizeof(float) * (brec.size() ? brec.size() : 4)));
    JAMD_HIP(hipMemcpy(g->d_book_rec, brec.data(), sizeof(float) * brec.size(), hipMemcpyHostToDevice));
    g->h_book_off = book_off;
    JAMD_HIP(hipMalloc(&g->d_book_off, sizeof(int) * (g->nbook + 1)));
    JAMD_HIP(hipMemcpy(g->d_book_off, book_off.data(), sizeof(int) * (g->nbook + 1), hipMemcpyHostToDevice));
    JAMD_HIP(hipMalloc(&g->d_st_book, sizeof(int) * g->S));
    JAMD_HIP(hipMemcpy(g->d_st_book, d->st_book, sizeof(int) * g->S, hipMemcpyHostToDevice));
    JAMD_HIP(hipMalloc(&g->d_ent_logw, sizeof(float) * (g->E ? g->E : 1)));
    JAMD_HIP(hipMemcpy(g->d_ent_logw, d->ent_logw, sizeof(float) * g->E, hipMemcpyHostToDevice));
    JAMD_HIP(hipMalloc(&g->d_tied_states, sizeof(int) * g->ntied));
    JAMD_HIP(hipMemcpy(g->d_tied_states, tied.data(), sizeof(int) * g->ntied, hipMemcpyHostToDevice));
  }
  return JAMD_OK;
}


int jamd_gmm_create(jamd_engine *e, const jamd_gmm_desc *d, int gprune, int gprune_num,
                    jamd_gmm **out) {
  if (!e || !d || !out) { jamd_set_error("jamd_gmm_create: NULL argument"); return JAMD_EINVAL; }
  *out = nullptr;
  jamd_gmm *g = nullptr;
  const int rc = gmm_create_impl(e, d, gprune, gprune_num, &g);
  if (rc != JAMD_OK) { if (g) jamd_gmm_destroy(g); return rc; }   // no leak on a failed allocation or a bad descriptor
  *out = g;
  return JAMD_OK;
}

void jamd_gmm_destroy(jamd_gmm *g) {
  if (!g) return;
  (void)hipSetDevice(g->eng->device);
  void *ptrs[] = { g->d_rec, g->d_cur_utt_off, g->d_st_off, g->d_st_off_plain, g->d_tied_states, g->d_st_book, g->d_book_off, g->d_book_rec,
                   g->d_ent_logw, g->d_frames, g->d_out, g->d_tm_score, g->d_tm_id, g->d_tm_num, g->d_narrow };
  for (void *p : ptrs) if (p) (void)hipFree(p);
  if (g->h_utt_off) (void)hipHostFree(g->h_utt_off);
  if (g->ev_utt_off) (void)hipEventDestroy(g->ev_utt_off);
  delete g;
}

int jamd_gmm_nstate(const jamd_gmm *g) { return g ? g->S : -1; }
int jamd_gmm_veclen(const jamd_gmm *g) { return g ? g->D : -1; }
const char *jamd_gmm_last_kernel(const jamd_gmm *g) { return g ? g->last_kernel : ""; }

// number of per-Gaussian score columns: the mixture entries of a plain model in state order, the
// codebook Gaussians of a tied-mixture model in codebook order; 0 for a model that mixes both
int jamd_gmm_nentry(const jamd_gmm *g) {
  if (!g) return 0;
  if (g->ntied == 0) return g->E;
  if (g->ntied == g->S && !g->h_book_off.empty()) return g->h_book_off[g->nbook];
  return 0;
}

int jamd_gmm_book_offsets(const jamd_gmm *g, int *off, int cap) {
  if (!g || !off || g->ntied != g->S || (int)g->h_book_off.size() != g->nbook + 1 || cap < g->nbook + 1) {
    jamd_set_error("jamd_gmm_book_offsets: not an all-tied-mixture model, or buffer too small"); return JAMD_EINVAL;
  }
  memcpy(off, g->h_book_off.data(), sizeof(int) * (size_t)(g->nbook + 1));
  return JAMD_OK;
}

int jamd_gmm_dens_dev(jamd_gmm *g, const float *dev_frames, int T, float *dev_out, void *stream) {
  if (!g || !dev_frames || !dev_out || T < 0) { jamd_set_error("jamd_gmm_dens_dev: bad argument"); return JAMD_EINVAL; }
  const int E = jamd_gmm_nentry(g);
  if (E <= 0) {
    jamd_set_error("jamd_gmm_dens_dev: models mixing plain and tied-mixture states have no single column order");
    return JAMD_EINVAL;
  }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(g->eng->device));
  hipStream_t st = jamd_stream(g->eng, stream);
  const size_t lds = sizeof(float) * ((size_t)g->D * 64 + 64 * 17);
  if (lds > 64 * 1024) { jamd_set_error("jamd_gmm_dens_dev: vector length %d too large", g->D); return JAMD_EINVAL; }
  hipLaunchKernelGGL(gmm_dens_kernel, dim3((T + 63) / 64, (E + kDensChunk - 1) / kDensChunk), dim3(64), lds, st,
                     g->ntied ? g->d_book_rec : g->d_rec, dev_frames, dev_out, T, E, g->D, g->rec);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { jamd_set_error("jamd_gmm_dens_dev: launch failed: %s", hipGetErrorString(le)); return JAMD_ELAUNCH; }
  return JAMD_OK;
}

int jamd_gmm_dens_host(jamd_gmm *g, const float *host_frames, int T, float *host_out) {
  if (!g || !host_frames || !host_out || T < 0) { jamd_set_error("jamd_gmm_dens_host: bad argument"); return JAMD_EINVAL; }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(g->eng->device));
  float *d_fr = nullptr, *d_out = nullptr;
  int rc = JAMD_OK;
  hipStream_t st = g->eng->stream;
  const int E = jamd_gmm_nentry(g);
  if (E <= 0) { jamd_set_error("jamd_gmm_dens_host: no single column order for this model"); return JAMD_EINVAL; }
  if (hipMalloc(&d_fr, sizeof(float) * (size_t)T * g->D) != hipSuccess ||
      hipMalloc(&d_out, sizeof(float) * (size_t)T * E) != hipSuccess) {
    jamd_set_error("jamd_gmm_dens_host: out of device memory"); rc = JAMD_ENOMEM;
  }
  if (rc == JAMD_OK && hipMemcpyAsync(d_fr, host_frames, sizeof(float) * (size_t)T * g->D, hipMemcpyHostToDevice, st) != hipSuccess) rc = JAMD_ENODEV;
  if (rc == JAMD_OK) rc = jamd_gmm_dens_dev(g, d_fr, T, d_out, st);
  if (rc == JAMD_OK && (hipMemcpyAsync(host_out, d_out, sizeof(float) * (size_t)T * E, hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess)) { jamd_set_error("jamd_gmm_dens_host: copy failed"); rc = JAMD_ELAUNCH; }
  if (d_fr) (void)hipFree(d_fr);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

// utterance boundaries of the running call, for the one scoring path that cares where an input begins
static int set_utterances(jamd_gmm *g, const int *utt_off, int nutt, hipStream_t st) {
  if (g->hist_method == 0) return JAMD_OK;
  if ((size_t)(nutt + 1) > g->utt_off_cap) {
    if (g->d_cur_utt_off) JAMD_HIP(hipFree(g->d_cur_utt_off));
    g->d_cur_utt_off = nullptr; g->utt_off_cap = 0;
    JAMD_HIP(hipMalloc(&g->d_cur_utt_off, sizeof(int) * ((size_t)nutt + 1)));
    g->utt_off_cap = (size_t)nutt + 1;
  }
  // utt_off is the caller's memory: staged in a PINNED buffer the model owns, so that the copy is truly asynchronous (a
  // pipelining host keeps its scoring stream free of host waits; from pageable memory the runtime would either block or
  // stage).  An event behind the copy guards the buffer: the next call on this model waits for it before it rewrites the
  // staging copy (normally long done) -- d_cur_utt_off itself is ordered by the stream.
  if (g->ev_utt_off) JAMD_HIP(hipEventSynchronize(g->ev_utt_off));
  else JAMD_HIP(hipEventCreateWithFlags(&g->ev_utt_off, hipEventDisableTiming));
  if ((size_t)(nutt + 1) > g->h_utt_off_cap) {
    if (g->h_utt_off) JAMD_HIP(hipHostFree(g->h_utt_off));
    g->h_utt_off = nullptr; g->h_utt_off_cap = 0;
    const size_t cap = (size_t)nutt + 1 < 1024 ? 1024 : (size_t)nutt + 1;
    JAMD_HIP(hipHostMalloc((void **)&g->h_utt_off, sizeof(int) * cap, hipHostMallocDefault));
    g->h_utt_off_cap = cap;
  }
  memcpy(g->h_utt_off, utt_off, sizeof(int) * ((size_t)nutt + 1));
  JAMD_HIP(hipMemcpyAsync(g->d_cur_utt_off, g->h_utt_off, sizeof(int) * ((size_t)nutt + 1), hipMemcpyHostToDevice, st));
  JAMD_HIP(hipEventRecord(g->ev_utt_off, st));
  g->cur_nutt = nutt;
  return JAMD_OK;
}

int jamd_gmm_outprob_dev(jamd_gmm *g, const float *dev_frames, int T, float *dev_out, void *stream) {
  const int off[2] = {0, T};
  if (T < 0) { jamd_set_error("jamd_gmm_outprob_dev: bad argument"); return JAMD_EINVAL; }
  return jamd_gmm_outprob_utts_dev(g, dev_frames, off, 1, dev_out, stream);
}

int jamd_gmm_outprob_utts_dev(jamd_gmm *g, const float *dev_frames, const int *utt_off, int nutt, float *dev_out, void *stream) {
  if (!g || !dev_frames || !dev_out || !utt_off || nutt < 1 || utt_off[0] != 0) {
    jamd_set_error("jamd_gmm_outprob_utts_dev: bad argument");
    return JAMD_EINVAL;
  }
  for (int u = 0; u < nutt; u++)
    if (utt_off[u + 1] < utt_off[u]) { jamd_set_error("jamd_gmm_outprob_utts_dev: utt_off must be non-decreasing"); return JAMD_EINVAL; }
  const int T = utt_off[nutt];
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(g->eng->device));
  hipStream_t st = jamd_stream(g->eng, stream);
  int rc = JAMD_OK;
  if ((rc = set_utterances(g, utt_off, nutt, st)) != JAMD_OK) return rc;
  if (g->E_plain == 0 && g->ntied == g->S) {
    // all states tied-mixture: nothing for the plain-state kernels to do
  } else if (g->gprune == JAMD_GPRUNE_SAFE && g->gprune_num < g->maxmix) {
    rc = jamd_gmm_launch_safe(g, dev_frames, T, dev_out, st);
  } else
  // gprune safe with N >= the largest mixture keeps every Gaussian but in
  // descending-score order (gprune_common.c:88); that order changes the
  // table log-sum, so it also goes through the sorted kernel
  if (g->gprune == JAMD_GPRUNE_SAFE) {
    rc = jamd_gmm_launch_safe(g, dev_frames, T, dev_out, st);
  } else
  if (T <= kNarrowT && narrow_enabled() && (g->D == 39 || g->D == 38 || g->D == 26 || g->D == 25)) {
    // a handful of frames: one lane per mixture entry instead of one lane per frame (K1n above)
    switch (g->D) {
      case 39: rc = launch_narrow<39>(g, dev_frames, T, dev_out, st); break;
      case 38: rc = launch_narrow<38>(g, dev_frames, T, dev_out, st); break;
      case 26: rc = launch_narrow<26>(g, dev_frames, T, dev_out, st); break;
      default: rc = launch_narrow<25>(g, dev_frames, T, dev_out, st); break;
    }
  } else
  switch (g->D) {
    case 39: rc = launch_tile<39, 2, 16>(g, dev_frames, T, dev_out, st); break;
    case 38: rc = launch_tile<38, 2, 16>(g, dev_frames, T, dev_out, st); break;
    case 26: rc = launch_tile<26, 2, 16>(g, dev_frames, T, dev_out, st); break;
    case 25: rc = launch_tile<25, 2, 16>(g, dev_frames, T, dev_out, st); break;
    default: rc = launch_tile_generic<2, 16>(g, dev_frames, T, dev_out, st); break;
  }
  if (rc != JAMD_OK) return rc;
  if (g->ntied) {
    // calc_tied_mix(): codebook top-N cache per (frame, book), then the states
    const size_t n = (size_t)T * g->nbook * g->tm_cap;
    if ((rc = ensure(&g->d_tm_score, &g->tm_cap_bytes, sizeof(float) * n)) != JAMD_OK) return rc;
    if ((rc = ensure((float **)&g->d_tm_id, &g->tm_id_bytes, sizeof(int) * n)) != JAMD_OK) return rc;
    if ((rc = ensure((float **)&g->d_tm_num, &g->tm_num_bytes, sizeof(int) * (size_t)T * g->nbook)) != JAMD_OK) return rc;
    if ((rc = jamd_gmm_launch_tmix(g, dev_frames, T, dev_out, g->d_tm_score, g->d_tm_id, g->d_tm_num, st)) != JAMD_OK) return rc;
  }
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

int jamd_gmm_tmix_cap(const jamd_gmm *g) { return g ? g->tm_cap : -1; }
int jamd_gmm_nbook(const jamd_gmm *g) { return g ? g->nbook : -1; }

int jamd_gmm_tmix_cache_dev(jamd_gmm *g, const float *dev_frames, int T, float *dev_score,
                            int *dev_id, int *dev_num, void *stream) {
  if (!g || !dev_frames || !dev_score || !dev_id || !dev_num || T < 0) {
    jamd_set_error("jamd_gmm_tmix_cache_dev: bad argument");
    return JAMD_EINVAL;
  }
  if (!g->ntied) { jamd_set_error("jamd_gmm_tmix_cache_dev: model has no tied-mixture states"); return JAMD_ESTATE; }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(g->eng->device));
  const int off[2] = {0, T};
  int rc = set_utterances(g, off, 1, jamd_stream(g->eng, stream));
  if (rc == JAMD_OK) rc = jamd_gmm_launch_tmix(g, dev_frames, T, nullptr, dev_score, dev_id, dev_num, jamd_stream(g->eng, stream));
  if (rc != JAMD_OK) return rc;
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    jamd_set_error("jamd_gmm_tmix_cache_dev: launch failed: %s", hipGetErrorString(le));
    return JAMD_ELAUNCH;
  }
  return JAMD_OK;
}

}  // extern "C"
