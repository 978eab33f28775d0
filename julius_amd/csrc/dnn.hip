// dnn.hip -- DNN-HMM state scores on gfx950 (K3/K4): the only place MFMA is used.
//
// Replaces dnn_calc_outprob() (libsent/src/phmm/calc_dnn.c:774-868) for a BATCH
// of frames (the reference is strictly frame-at-a-time, 00readme-DNN.txt:29-34):
//   per layer   dst = W src + b         calc_dnn_fma.c:19-80 (the SIMD path the
//                                        reference selects on an FMA host)
//   hidden      table logistic          calc_dnn.c:342-369, :813-818
//   output      x_i -> INV_LOG_TEN*(x_i - addlog_array(x)) - state_prior[i]
//                                        calc_dnn.c:858-866, addlog.c:103-123
//
// Numerical contract: bit-exact with the reference's FMA kernel.  That kernel
// keeps EIGHT partial sums per output (AVX lanes l = k mod 8), each a fused
// multiply-add chain over k = l, l+8, l+16, ..., and finally adds lanes 0..7
// left to right and then the bias (calc_dnn_fma.c:53-60).
// v_mfma_f32_32x32x2_f32 is, per output element, exactly a k-ordered fmaf chain
// (cdna_hip_programming.md section 3), so each 32x32 output tile keeps EIGHT
// MFMA accumulators, accumulator l being fed only k == l (mod 8) in ascending
// order (lanes 0-31 supply k, lanes 32-63 supply k+8), and the epilogue adds the
// eight accumulators left to right, then the bias.  Same MFMA count as a single
// accumulator; the price is 128 accumulator registers per wave tile.
//
// GEMM shape: C[t][o] = sum_k X[t][k] * W[o][k]  (both operands K-contiguous).
// Block = 4 waves = 64 frames x 64 outputs, K slab 64 staged through a DOUBLE-
// BUFFERED LDS tile filled by LDS-DMA (global_load_lds_dwordx4, one barrier per
// slab: the DMA of the next slab is issued before the MFMAs of the current one
// and lands in the other buffer); rows are unpadded and quad-swizzled (quad c of
// row r at position c ^ (r & 15)) so that the ds_read_b128 fragment reads (one
// 16-byte quad of k per lane) are conflict free for the instruction's 16-lane
// groups (MI355X_MICROARCH.md, LDS).
#include "jamd_device.h"

struct jamd_dnn {
  jamd_engine *eng = nullptr;
  int nlayer = 0;
  std::vector<int> dims;
  std::vector<float *> d_w, d_b;   // W[l]: [dims[l+1]][dims[l]] as given
  float *d_prior = nullptr;
  float *d_zero = nullptr;           // 64 zero bytes: DMA source of out-of-range quads
  int maxdim = 0;
  float *d_act[2] = {nullptr, nullptr}; size_t act_cap = 0;
  float *d_lse = nullptr; size_t lse_cap = 0;
  float *d_frames = nullptr; size_t frames_cap = 0;
  float *d_out = nullptr; size_t out_cap = 0;
  hipStream_t side = nullptr;          // softmax tail of chunk c runs here while chunk c+1's GEMMs run on the caller's stream
  hipEvent_t ev_gemm[8] = {}, ev_tail = nullptr, ev_start = nullptr;
};

namespace {
using namespace jamd;

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, KS = 64;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// ACT: 1 = table logistic (hidden layer), 0 = raw (output layer).
// zero16: 16 bytes of zeros in global memory, the source of every out-of-range quad (zero
// padding keeps every chain exact: fma(0,0,acc) == acc).
template <int ACT>
__global__ void __launch_bounds__(256, 2)
dnn_layer_kernel(const float *__restrict__ X, const float *__restrict__ W,
                 const float *__restrict__ bias, const float *__restrict__ sig,
                 float *__restrict__ Y, int T, int K, int N, int ldx, int ldy, int nmb,
                 const float *__restrict__ zero16) {
  // K slab of the X tile and of the W tile, two buffers each, UNPADDED rows of 16 quads: the
  // tiles arrive by LDS-DMA (global_load_lds_dwordx4: a wave writes 1 KB = 4 rows lane-linearly,
  // no VGPR round trip, no ds_write), so padding is impossible; instead quad c of row r is kept
  // at position c ^ (r & 15) -- the permutation is applied to the per-lane SOURCE address --
  // which makes the ds_read_b128 fragment reads of 16 consecutive rows conflict free.
  __shared__ __align__(16) float Xs[2][BM][KS];
  __shared__ __align__(16) float Ws[2][BN][KS];
  const int nnb = (N + BN - 1) / BN;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  // The dispatcher puts block b on XCD b % 8.  Each XCD owns every eighth 64-output tile for ALL
  // frame strips: its share of W (1/8 of the layer, 2 MB at 2048 x 2048) stays in its 4 MB L2
  // while the strips of X stream through -- instead of all of W streaming through every L2
  // once per strip (17 GB -> 5 GB of L2 fills per hidden layer at 64 000 frames).
  const int tpx = (nnb + 7) / 8;
  const int nb = xcd + 8 * (q % tpx), mb = q / tpx;
  if (mb >= nmb || nb >= nnb) return;
  const int t0 = mb * BM, o0 = nb * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

  // staging: 32 DMA instructions per slab (16 for X, 16 for W), eight per wave.  Instruction
  // id = 8 * wave + j moves rows 4 * (id & 15) .. + 3 of X (id < 16) or W; lane = 16 * (row in
  // group) + position, and fetches the quad that belongs at that position.
  constexpr int NI = 8;
  const float *src[NI];     // row base + 4 * quad (or the zero block), K offset added per slab
  int cq[NI];               // 4 * logical quad (k offset inside the slab) of this lane
  int kmask[NI];            // ~0, or 0 for a W row past N: the source stays on the zero block
#pragma unroll
  for (int j = 0; j < NI; j++) {
    const int id = 8 * wave + j, r = 4 * (id & 15) + (lane >> 4);
    const int c = (lane & 15) ^ (r & 15);
    cq[j] = 4 * c;
    if (id < 16) {
      int tr = t0 + r; if (tr > T - 1) tr = T - 1;
      src[j] = X + (size_t)tr * ldx + 4 * c; kmask[j] = ~0;
    } else {
      const int orow = o0 + r;
      kmask[j] = orow < N ? ~0 : 0;
      src[j] = orow < N ? W + (size_t)orow * K + 4 * c : zero16;
    }
  }
  // K is a multiple of 8 (jamd_dnn_create), so a quad is either fully inside a row or fully
  // past its end
  // a slab that lies completely inside K needs no per-quad test (all but the last slab of a
  // layer whose K is not a multiple of 64)
  auto stage = [&](int buf, int k0) {
    const bool full = k0 + KS <= K;
#pragma unroll
    for (int j = 0; j < NI; j++) {
      const int id = 8 * wave + j;
      const float *g = src[j] + (k0 & kmask[j]);
      if (!full && !(k0 + cq[j] < K)) g = zero16;
      float *dst = (id < 16) ? &Xs[buf][4 * (id & 15)][0] : &Ws[buf][4 * (id & 15)][0];
      __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)dst, 16, 0, 0);
    }
  };

  f16v acc[8];
#pragma unroll
  for (int l = 0; l < 8; l++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[l][r] = 0.0f;

  stage(0, 0);
  __syncthreads();          // carries the vmcnt(0) that lands the DMA
  const int arow = wm + (lane & 31), brow = wn + (lane & 31), half = lane >> 5;
  const int asw = arow & 15, bsw = brow & 15;
  int cur = 0;
  for (int k0 = 0; k0 < K; k0 += KS) {
    if (k0 + KS < K) stage(cur ^ 1, k0 + KS);   // the other buffer was last read before the previous barrier
#pragma unroll
    for (int g = 0; g < KS; g += 16) {
      // lanes 0-31 take k = g+0..7, lanes 32-63 k = g+8..15: accumulator l sees
      // k = g+l then g+8+l -- ascending within its residue class mod 8
      const int q0 = g / 4 + 2 * half;
      const f4v a0 = *(const f4v *)&Xs[cur][arow][4 * (q0 ^ asw)];
      const f4v a1 = *(const f4v *)&Xs[cur][arow][4 * ((q0 + 1) ^ asw)];
      const f4v b0 = *(const f4v *)&Ws[cur][brow][4 * (q0 ^ bsw)];
      const f4v b1 = *(const f4v *)&Ws[cur][brow][4 * ((q0 + 1) ^ bsw)];
#pragma unroll
      for (int l = 0; l < 4; l++) {
        acc[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[l], b0[l], acc[l], 0, 0, 0);
        acc[l + 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[l], b1[l], acc[l + 4], 0, 0, 0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: lanes add 0..7 left to right, then the bias (calc_dnn_fma.c:53-60)
  const int j = o0 + wn + (lane & 31);
  const float bj = (j < N) ? bias[j] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int t = t0 + wm + i;
    float s = acc[0][r] + acc[1][r];
#pragma unroll
    for (int l = 2; l < 8; l++) s = s + acc[l][r];
    s = s + bj;
    if (ACT) {
      // calc_dnn.c:813-818: clamp at +-8, else table[(int)((x + 8.0f) * 20000 + 0.5)]
      float y;
      if (s <= -8.0f) y = (float)0.000334;
      else if (s >= 8.0f) y = (float)0.999666;
      else y = sig[(int)((double)((s + 8.0f) * (float)JAMD_LOGISTIC_FACTOR) + 0.5)];
      s = y;
    }
    if (t < T && j < N) Y[(size_t)t * ldy + j] = s;
  }
}

// Output layer, step 1: logprob = addlog_array(x, S) (calc_dnn.c:862), the
// right-to-left table scan; inherently serial per frame, one lane per frame.
__global__ void __launch_bounds__(64)
dnn_lse_kernel(const float *__restrict__ x, const float *__restrict__ tbl, float *__restrict__ lse,
               int T, int S, float addmin_f) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= T) return;
  const float *row = x + (size_t)t * S;
  float y = JAMD_LOG_ZERO;
  for (int n = S - 1; n >= 0; n--) y = addlog_step(y, row[n], tbl, addmin_f);
  lse[t] = y;
}

// step 2: last_cache[i] = INV_LOG_TEN * (x_i - logprob) - state_prior[i]
// (double expression rounded once on store, calc_dnn.c:864)
__global__ void __launch_bounds__(256)
dnn_norm_kernel(float *__restrict__ x, const float *__restrict__ lse, const float *__restrict__ prior,
                int T, int S) {
  // blockIdx.y walks frames, the block's threads walk the states of a frame four at a time
  // (S is a multiple of 4 whenever the rows are 16-byte aligned; a scalar tail covers the rest)
  const int s4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (s4 >= S) return;
  const bool vec = (S & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(prior) & 15) == 0;
  for (int t = blockIdx.y; t < T; t += gridDim.y) {
    float *row = x + (size_t)t * S;
    if (vec) {
      float4 v = *reinterpret_cast<const float4 *>(row + s4);
      const float4 p = *reinterpret_cast<const float4 *>(prior + s4);
      const float lf = lse[t];
      v.x = (float)(JAMD_INV_LOG_TEN * (double)(v.x - lf) - (double)p.x);
      v.y = (float)(JAMD_INV_LOG_TEN * (double)(v.y - lf) - (double)p.y);
      v.z = (float)(JAMD_INV_LOG_TEN * (double)(v.z - lf) - (double)p.z);
      v.w = (float)(JAMD_INV_LOG_TEN * (double)(v.w - lf) - (double)p.w);
      *reinterpret_cast<float4 *>(row + s4) = v;
    } else {
      const float lf = lse[t];
      for (int s = s4; s < S && s < s4 + 4; s++)
        row[s] = (float)(JAMD_INV_LOG_TEN * (double)(row[s] - lf) - (double)prior[s]);
    }
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

int jamd_dnn_create(jamd_engine *e, const jamd_dnn_desc *d, jamd_dnn **out) {
  if (!e || !d || !out || d->nlayer < 1 || !d->dims || !d->w || !d->b || !d->state_prior) {
    jamd_set_error("jamd_dnn_create: bad argument");
    return JAMD_EINVAL;
  }
  *out = nullptr;
  for (int l = 0; l <= d->nlayer; l++) {
    if (d->dims[l] <= 0) { jamd_set_error("jamd_dnn_create: dims[%d]=%d", l, d->dims[l]); return JAMD_EINVAL; }
    // dnn_layer_load() insists on 8-element aligned inputs on AVX/FMA hosts
    // (calc_dnn.c:395) and calc_dnn_fma() would silently drop a tail
    if (l < d->nlayer && d->dims[l] % 8 != 0) {
      jamd_set_error("jamd_dnn_create: layer %d input length %d is not a multiple of 8 "
                     "(same restriction as the reference's SIMD path)", l, d->dims[l]);
      return JAMD_EINVAL;
    }
  }
  JAMD_HIP(hipSetDevice(e->device));
  jamd_dnn *n = new jamd_dnn();
  n->eng = e; n->nlayer = d->nlayer;
  n->dims.assign(d->dims, d->dims + d->nlayer + 1);
  for (int v : n->dims) if (v > n->maxdim) n->maxdim = v;
  for (int l = 0; l < d->nlayer; l++) {
    float *w = nullptr, *b = nullptr;
    const size_t nw = (size_t)n->dims[l + 1] * n->dims[l];
    JAMD_HIP(hipMalloc(&w, sizeof(float) * nw));
    JAMD_HIP(hipMemcpy(w, d->w[l], sizeof(float) * nw, hipMemcpyHostToDevice));
    JAMD_HIP(hipMalloc(&b, sizeof(float) * n->dims[l + 1]));
    JAMD_HIP(hipMemcpy(b, d->b[l], sizeof(float) * n->dims[l + 1], hipMemcpyHostToDevice));
    n->d_w.push_back(w); n->d_b.push_back(b);
  }
  const int S = n->dims[d->nlayer];
  JAMD_HIP(hipMalloc(&n->d_prior, sizeof(float) * S));
  JAMD_HIP(hipMemcpy(n->d_prior, d->state_prior, sizeof(float) * S, hipMemcpyHostToDevice));
  JAMD_HIP(hipMalloc(&n->d_zero, 64));
  JAMD_HIP(hipMemset(n->d_zero, 0, 64));
  *out = n;
  return JAMD_OK;
}

void jamd_dnn_destroy(jamd_dnn *n) {
  if (!n) return;
  (void)hipSetDevice(n->eng->device);
  for (float *p : n->d_w) (void)hipFree(p);
  for (float *p : n->d_b) (void)hipFree(p);
  if (n->side) {
    (void)hipStreamSynchronize(n->side);
    for (int i = 0; i < 8; i++) (void)hipEventDestroy(n->ev_gemm[i]);
    (void)hipEventDestroy(n->ev_tail); (void)hipEventDestroy(n->ev_start);
    (void)hipStreamDestroy(n->side);
  }
  float *ptrs[] = { n->d_prior, n->d_zero, n->d_act[0], n->d_act[1], n->d_lse, n->d_frames, n->d_out };
  for (float *p : ptrs) if (p) (void)hipFree(p);
  delete n;
}

int jamd_dnn_nstate(const jamd_dnn *n) { return n ? n->dims[n->nlayer] : -1; }
int jamd_dnn_veclen(const jamd_dnn *n) { return n ? n->dims[0] : -1; }

int jamd_dnn_outprob_dev(jamd_dnn *n, const float *dev_frames, int T, float *dev_out, void *stream) {
  if (!n || !dev_frames || !dev_out || T < 0) {
    jamd_set_error("jamd_dnn_outprob_dev: bad argument");
    return JAMD_EINVAL;
  }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(n->eng->device));
  hipStream_t st = jamd_stream(n->eng, stream);
  int rc;
  // hidden activations ping-pong between two [T][maxhidden] buffers
  int maxh = 1;
  for (int l = 1; l < n->nlayer; l++) if (n->dims[l] > maxh) maxh = n->dims[l];
  const size_t need = sizeof(float) * (size_t)T * maxh;
  if (n->act_cap < need) {
    for (int k = 0; k < 2; k++) { if (n->d_act[k]) JAMD_HIP(hipFree(n->d_act[k])); n->d_act[k] = nullptr; }
    n->act_cap = 0;
    JAMD_HIP(hipMalloc(&n->d_act[0], need));
    JAMD_HIP(hipMalloc(&n->d_act[1], need));
    n->act_cap = need;
  }
  if ((rc = ensure(&n->d_lse, &n->lse_cap, sizeof(float) * (size_t)T)) != JAMD_OK) return rc;
  // The output layer is followed by the serial row log-sum (dnn_lse: one lane per frame, a
  // latency-bound chain of table gathers that keeps ~1 wave per CU busy) and the
  // normalisation.  For very long batches (>= 131072 frames) the frames are cut into up to 8 chunks of 32768+ frames; chunk c's tail
  // runs on a side stream and overlaps chunk c+1's GEMMs, whose blocks leave registers and
  // issue slots free for it.
  const int S = n->dims[n->nlayer];
  int nchunk = T >= 131072 ? (T + 32767) / 32768 : 1;  // measured: pays (+4.5 %) only for very long batches
  if (nchunk > 8) nchunk = 8;
#ifdef JAMD_DEV
  if (getenv("JAMD_DNN_NOCHUNK")) nchunk = 1;
#endif
  int per = ((T + nchunk - 1) / nchunk + BM - 1) / BM * BM;
  if (nchunk > 1 && n->side == nullptr) {
    JAMD_HIP(hipStreamCreateWithFlags(&n->side, hipStreamNonBlocking));
    for (int i = 0; i < 8; i++) JAMD_HIP(hipEventCreateWithFlags(&n->ev_gemm[i], hipEventDisableTiming));
    JAMD_HIP(hipEventCreateWithFlags(&n->ev_tail, hipEventDisableTiming));
    JAMD_HIP(hipEventCreateWithFlags(&n->ev_start, hipEventDisableTiming));
  }
  if (nchunk > 1) {     // the side stream must not start before earlier work on the caller's stream
    JAMD_HIP(hipEventRecord(n->ev_start, st));
    JAMD_HIP(hipStreamWaitEvent(n->side, n->ev_start, 0));
  }
  for (int c = 0, t0 = 0; t0 < T; c++, t0 += per) {
    const int Tc = (T - t0 < per) ? T - t0 : per;
    const int nmb = (Tc + BM - 1) / BM;
    const float *src = dev_frames + (size_t)t0 * n->dims[0];
    for (int l = 0; l < n->nlayer; l++) {
      const int K = n->dims[l], N = n->dims[l + 1];
      const bool last = (l == n->nlayer - 1);
      float *dst = last ? dev_out + (size_t)t0 * S : n->d_act[l & 1];
      const int nnb = (N + BN - 1) / BN;
      const int grid = 8 * ((nnb + 7) / 8) * nmb;
      if (last)
        hipLaunchKernelGGL((dnn_layer_kernel<0>), dim3(grid), dim3(256), 0, st, src, n->d_w[l], n->d_b[l],
                           n->eng->d_logistic, dst, Tc, K, N, K, N, nmb, n->d_zero);
      else
        hipLaunchKernelGGL((dnn_layer_kernel<1>), dim3(grid), dim3(256), 0, st, src, n->d_w[l], n->d_b[l],
                           n->eng->d_logistic, dst, Tc, K, N, K, N, nmb, n->d_zero);
      src = dst;
    }
    hipStream_t ts = st;
    if (nchunk > 1) {
      JAMD_HIP(hipEventRecord(n->ev_gemm[c], st));
      JAMD_HIP(hipStreamWaitEvent(n->side, n->ev_gemm[c], 0));
      ts = n->side;
    }
    float *oc = dev_out + (size_t)t0 * S;
    hipLaunchKernelGGL(dnn_lse_kernel, dim3((Tc + 63) / 64), dim3(64), 0, ts, oc, n->eng->d_addlog,
                       n->d_lse + t0, Tc, S, n->eng->addmin_f);
    hipLaunchKernelGGL(dnn_norm_kernel, dim3((S + 1023) / 1024, Tc < 4096 ? Tc : 4096), dim3(256), 0, ts, oc,
                       n->d_lse + t0, n->d_prior, Tc, S);
  }
  if (nchunk > 1) {     // join: the caller's stream continues only after the last tail
    JAMD_HIP(hipEventRecord(n->ev_tail, n->side));
    JAMD_HIP(hipStreamWaitEvent(st, n->ev_tail, 0));
  }
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    jamd_set_error("jamd_dnn_outprob_dev: launch failed: %s", hipGetErrorString(le));
    return JAMD_ELAUNCH;
  }
  return JAMD_OK;
}

int jamd_dnn_outprob_host(jamd_dnn *n, const float *host_frames, int T, float *host_out) {
  if (!n || !host_frames || !host_out || T < 0) {
    jamd_set_error("jamd_dnn_outprob_host: bad argument");
    return JAMD_EINVAL;
  }
  if (T == 0) return JAMD_OK;
  JAMD_HIP(hipSetDevice(n->eng->device));
  const int D = n->dims[0], S = n->dims[n->nlayer];
  int rc;
  if ((rc = ensure(&n->d_frames, &n->frames_cap, sizeof(float) * (size_t)T * D)) != JAMD_OK) return rc;
  if ((rc = ensure(&n->d_out, &n->out_cap, sizeof(float) * (size_t)T * S)) != JAMD_OK) return rc;
  hipStream_t st = n->eng->stream;
  JAMD_HIP(hipMemcpyAsync(n->d_frames, host_frames, sizeof(float) * (size_t)T * D, hipMemcpyHostToDevice, st));
  if ((rc = jamd_dnn_outprob_dev(n, n->d_frames, T, n->d_out, st)) != JAMD_OK) return rc;
  JAMD_HIP(hipMemcpyAsync(host_out, n->d_out, sizeof(float) * (size_t)T * S, hipMemcpyDeviceToHost, st));
  hipError_t se = hipStreamSynchronize(st);
  if (se != hipSuccess) {
    jamd_set_error("jamd_dnn_outprob_host: execution failed: %s", hipGetErrorString(se));
    return JAMD_ELAUNCH;
  }
  return JAMD_OK;
}

}  // extern "C"
