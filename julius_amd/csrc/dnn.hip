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
// (cdna_hip_programming.md section 3; lanes 0-31 supply k, lanes 32-63 the next
// k of the chain), so a chain is one MFMA accumulator fed the k of its residue
// class in ascending order.  The eight chains are independent; they run ONE
// AFTER THE OTHER over all of K and are folded into the sum in the reference's
// order (dnn_layer_rs_kernel below) -- two accumulators per 32x32 tile.  (The
// first version kept all eight side by side: 128 accumulator registers per
// 32x32 wave tile, no fragment reuse, 104 TFLOP/s; this form: 110.)
//
// GEMM shape: C[t][o] = sum_k X[t][k] * W[o][k], operands in residue-major
// rows.  Block = 4 waves = 128 frames x 128 outputs, wave tile 64 x 64, slabs
// of 32 chain entries staged through a DOUBLE-BUFFERED LDS tile filled by
// LDS-DMA (global_load_lds_dwordx4, one barrier per slab: the DMA of the next
// slab is issued before the MFMAs of the current one and lands in the other
// buffer); rows are unpadded and quad-swizzled so that the ds_read_b128
// fragment reads are conflict free for the instruction's 16-lane groups
// (MI355X_MICROARCH.md, LDS).
#include "jamd_device.h"

struct jamd_dnn {
  jamd_engine *eng = nullptr;
  int nlayer = 0;
  std::vector<int> dims;
  std::vector<float *> d_b;
  std::vector<float *> d_wr;       // W[l] in residue-major rows (see dnn_layer_rs_kernel), [dims[l+1]][8 * kmp[l]]
  std::vector<int> kmp;            // per layer input: padded length of one residue segment = ceil(dims[l] / 8 / 8) * 8
  float *d_xr = nullptr; size_t xr_cap = 0;   // the frames in residue-major rows
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

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// ---------------------------------------------------------------------------------------------------------
// Residue-serial form of the layer.  The reference's eight partial sums per output are eight INDEPENDENT fused
// multiply-add chains (k = l, l+8, l+16, ...) that are added left to right at the end.  Nothing obliges us to
// run them side by side: run chain 0 over all of K, then chain 1, ... and fold each finished chain into a running
// sum S = (((a0 + a1) + a2) + ...) -- the same floats, with TWO accumulators per 32x32 tile instead of eight.
// The registers that frees buy a 64x64 wave tile (2x2 MFMA tiles) and a 128x128 block tile: every LDS fragment is
// used by two MFMAs instead of one and a byte fetched into LDS feeds twice the flops, i.e. half the fragment
// reads, half the LDS-DMA pieces and half the barriers per MFMA of a kernel that keeps the eight chains side by side.
//
// Operands are read in "residue-major" rows: row r holds, for l = 0..7, the segment {x[8m + l]: m = 0..K/8-1}
// padded with zeros to kmp = a multiple of 8 entries; inside each group of eight m the even ones come first
// (m = 0,2,4,6,1,3,5,7), so that one 16-byte quad is the lower (or upper) half-wave's operand of four consecutive
// MFMAs (lanes 0-31 supply k, lanes 32-63 the next k of the chain).  The weights are packed once at creation, the
// frames by dnn_pack_rm_kernel, and a hidden layer's epilogue writes its activations directly in this form.
//
// LDS tile: 128 rows x 32 floats per operand and buffer; quad c of row r is kept at position c ^ ((r >> 1) & 7):
// with 128-byte rows the 16 lanes a ds_read_b128 services per cycle ({0-3,12-15,20-27} etc., MI355X_MICROARCH.md)
// then hit 16 different 16-byte slots of the 256-byte bank window.
constexpr int RB = 128;                  // block tile: frames and outputs
constexpr int MS = 32;                   // entries of one residue segment per slab

__device__ __forceinline__ int rm_pos(int m) {           // place of entry m inside its residue segment
  const int e = m & 7;
  return (m & ~7) | ((e & 1) ? 4 + (e >> 1) : (e >> 1));
}

// frames [T][K] (row stride ldx) -> residue-major rows [T][8 * kmp]
__global__ void __launch_bounds__(256)
dnn_pack_rm_kernel(const float *__restrict__ X, float *__restrict__ Xr, int T, int K, int ldx, int kmp) {
  const int L = 8 * kmp;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < L; idx += gridDim.x * blockDim.x) {
    const int l = idx / kmp, p = idx - l * kmp;
    const int e = p & 7, m = (p & ~7) | (e < 4 ? 2 * e : 2 * (e - 4) + 1);
    const bool in = 8 * m + l < K;
    for (int t = blockIdx.y; t < T; t += gridDim.y)
      Xr[(size_t)t * L + idx] = in ? X[(size_t)t * ldx + 8 * m + l] : 0.0f;
  }
}

// OUT: 0 = output layer (raw values, natural order, row stride ldy), 1 = hidden layer (table logistic, written as
// residue-major rows of 8 * kmp_out entries, zero padded)
// STRADDLE: the residue segments are not a whole number of slabs long (kmp % 32 != 0: the 528-wide first layer has
// 72-entry segments).  The chains then do not get slabs of their own -- three per chain, the third a quarter full:
// 24 slabs of staging, barriers and pipeline fill for 18 slabs of data -- but the operand rows are streamed as they
// lie (the eight segments are contiguous) in 8 * kmp / 32 slabs, and a chain is folded into the running sum in the
// middle of a slab, after its last group of eight entries.  Same multiplications in the same chains.
template <int OUT, bool STRADDLE>
__global__ void __launch_bounds__(256, 2)
dnn_layer_rs_kernel(const float *__restrict__ Xr, const float *__restrict__ Wr, const float *__restrict__ bias,
                    const float *__restrict__ sig, float *__restrict__ Y, int T, int kmp, int N, int ncover,
                    int ldy, int kmp_out, int nmb, const float *__restrict__ zero16) {
  __shared__ __align__(16) float Xs[2][RB * MS];
  __shared__ __align__(16) float Ws[2][RB * MS];
  const int L = 8 * kmp;                                  // operand row length
  const int nnb = (ncover + RB - 1) / RB;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  const int tpx = (nnb + 7) / 8;                          // each XCD owns every eighth output tile for all frame strips
  const int nb = xcd + 8 * (q % tpx), mb = q / tpx;
  if (mb >= nmb || nb >= nnb) return;
  const int t0 = mb * RB, o0 = nb * RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // staging: 32 DMA pieces per slab (16 per operand: eight 128-byte rows each), eight per wave.  Lane = 8 * (row in
  // piece) + position, and fetches the quad that belongs at that position.
  constexpr int NI = 8;
  const float *src[NI];                                   // row base + 4 * logical quad, or the zero block
  int cq[NI];                                             // 4 * logical quad (offset inside the slab)
  bool live[NI];
#pragma unroll
  for (int j = 0; j < NI; j++) {
    const int id = 8 * wave + j, r = 8 * (id & 15) + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    cq[j] = 4 * c;
    if (id < 16) {
      int tr = t0 + r; if (tr > T - 1) tr = T - 1;
      src[j] = Xr + (size_t)tr * L + 4 * c; live[j] = true;
    } else {
      const int orow = o0 + r;
      live[j] = orow < N;
      src[j] = live[j] ? Wr + (size_t)orow * L + 4 * c : zero16;
    }
  }
  auto stage = [&](int buf, int l, int m0) {                // STRADDLE: l = 0, m0 = offset in the whole row of L entries
    const int lim = STRADDLE ? L : kmp;
    const bool full = m0 + MS <= lim;
    const int off = l * kmp + m0;
#pragma unroll
    for (int j = 0; j < NI; j++) {
      const int id = 8 * wave + j;
      const float *g = live[j] ? src[j] + off : zero16;
      if (!full && !(m0 + cq[j] < lim)) g = zero16;
      float *dst = (id < 16) ? &Xs[buf][256 * (id & 15)] : &Ws[buf][256 * (id & 15)];
      __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)dst, 16, 0, 0);
    }
  };

  f16v S[2][2], acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) { S[i][j][r] = 0.0f; acc[i][j][r] = 0.0f; }

  const int half = lane >> 5, rr = lane & 31;
  int arow[2], brow[2], asw[2], bsw[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    arow[i] = wm + 32 * i + rr; brow[i] = wn + 32 * i + rr;
    asw[i] = (arow[i] >> 1) & 7; bsw[i] = (brow[i] >> 1) & 7;
  }
  const int nslab = (kmp + MS - 1) / MS;

  stage(0, 0, 0);
  __syncthreads();          // carries the vmcnt(0) that lands the DMA
  int cur = 0;
  if constexpr (STRADDLE) {
    const int gpc = kmp >> 3, ngroup = 8 * gpc, nslab_row = (L + MS - 1) / MS;   // groups per chain, in the row; slabs of the row
    int left = gpc;                                       // groups of the running chain still to come
    bool first = true;
    for (int s = 0; s < nslab_row; s++) {
      if (s + 1 < nslab_row) stage(cur ^ 1, 0, (s + 1) * MS);
#pragma unroll
      for (int g = 0; g < MS / 8; g++) {
        if (4 * s + g >= ngroup) break;
        const int c = 2 * g + half;
        f4v a[2], bq[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          a[i] = *(const f4v *)&Xs[cur][arow[i] * MS + 4 * (c ^ asw[i])];
          bq[i] = *(const f4v *)&Ws[cur][brow[i] * MS + 4 * (c ^ bsw[i])];
        }
#pragma unroll
        for (int st = 0; st < 4; st++)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][st], bq[j][st], acc[i][j], 0, 0, 0);
        if (--left == 0) {                                // the chain is complete (calc_dnn_fma.c:53-60: lanes added left to right)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
              for (int r = 0; r < 16; r++) { S[i][j][r] = first ? acc[i][j][r] : S[i][j][r] + acc[i][j][r]; acc[i][j][r] = 0.0f; }
          first = false; left = gpc;
        }
      }
      __syncthreads();
      cur ^= 1;
    }
  } else
  for (int l = 0; l < 8; l++) {
    for (int s = 0; s < nslab; s++) {
      {                                                   // the next slab: of this chain, or the first of the next one
        const int ln = (s + 1 < nslab) ? l : l + 1, sn = (s + 1 < nslab) ? s + 1 : 0;
        if (ln < 8) stage(cur ^ 1, ln, sn * MS);
      }
      const int ng = (kmp - s * MS) >> 3;                 // groups of eight entries this slab holds (the last may be short)
#pragma unroll
      for (int g = 0; g < MS / 8; g++) {
        if (g >= ng) break;
        // lanes 0-31 take the even entries 8g+{0,2,4,6}, lanes 32-63 the odd ones: MFMA step i multiplies entry
        // 8g+2i then 8g+2i+1 -- ascending along the chain
        const int c = 2 * g + half;
        f4v a[2], bq[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          a[i] = *(const f4v *)&Xs[cur][arow[i] * MS + 4 * (c ^ asw[i])];
          bq[i] = *(const f4v *)&Ws[cur][brow[i] * MS + 4 * (c ^ bsw[i])];
        }
#pragma unroll
        for (int st = 0; st < 4; st++)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][st], bq[j][st], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      cur ^= 1;
    }
    // chain l is complete: S = a0, then S + a1, ... (calc_dnn_fma.c:53-60 adds the lanes left to right)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) { S[i][j][r] = (l == 0) ? acc[i][j][r] : S[i][j][r] + acc[i][j][r]; acc[i][j][r] = 0.0f; }
  }

  // epilogue: + bias, activation, store
#pragma unroll
  for (int jt = 0; jt < 2; jt++) {
    const int j = o0 + wn + 32 * jt + rr;
    const float bj = (j < N) ? bias[j] : 0.0f;
    const int jpos = OUT ? (j & 7) * kmp_out + rm_pos(j >> 3) : j;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int t = t0 + wm + 32 * i + row;
        float v = S[i][jt][r] + bj;
        if (OUT) {
          // calc_dnn.c:813-818: clamp at +-8, else table[(int)((x + 8.0f) * 20000 + 0.5)]
          float y;
          if (v <= -8.0f) y = (float)0.000334;
          else if (v >= 8.0f) y = (float)0.999666;
          else y = sig[(int)((double)((v + 8.0f) * (float)JAMD_LOGISTIC_FACTOR) + 0.5)];
          v = (j < N) ? y : 0.0f;
          if (t < T && j < 8 * kmp_out) Y[(size_t)t * ldy + jpos] = v;
        } else {
          if (t < T && j < N) Y[(size_t)t * ldy + jpos] = v;
        }
      }
    }
  }
}

// Output layer, step 1: logprob = addlog_array(x, S) (calc_dnn.c:862), the
// right-to-left table scan; inherently serial per frame, one lane per frame.
// (A wave per frame that runs the scan only over the terms within -LOG_ADDMIN of the maximum to their right --
// the others provably leave the sum unchanged -- is bit-exact too, but 4x slower on a network whose outputs lie
// close together, as the random-init benchmark network's do: every term then takes the serial step.)
// (Round 3: the kernel is bound by the THROUGHPUT of the scattered table gathers -- 2.56e8 of them, each its own
// cache line, 190 G gathers/s -- not by their latency: spreading the frames over 2x..16x more waves, 32 down to 4
// frames a wave, changes nothing or loses: 34.85 / 34.97 / 35.38 / 37.09 ms for the whole network against 34.99.)
__global__ void __launch_bounds__(64)
dnn_lse_kernel(const float *__restrict__ x, const float *__restrict__ tbl, float *__restrict__ lse,
               int T, int S, float addmin_f) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= T) return;
  const float *row = x + (size_t)t * S;
  float y = JAMD_LOG_ZERO;
  int n = S - 1;
  // the step's latency is the dependent table gather; the row values do not depend on it: sixteen at a time, ahead
  for (; n >= 15; n -= 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = row[n - j];
#pragma unroll
    for (int j = 0; j < 16; j++) y = addlog_step(y, v[j], tbl, addmin_f);
  }
  for (; n >= 0; n--) y = addlog_step(y, row[n], tbl, addmin_f);
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
    float *b = nullptr;
    JAMD_HIP(hipMalloc(&b, sizeof(float) * n->dims[l + 1]));
    JAMD_HIP(hipMemcpy(b, d->b[l], sizeof(float) * n->dims[l + 1], hipMemcpyHostToDevice));
    n->d_b.push_back(b);
    // the weights as residue-major rows (dnn_layer_rs_kernel)
    const int K = n->dims[l], N = n->dims[l + 1], kmp = ((K / 8 + 7) / 8) * 8;
    std::vector<float> wr((size_t)N * 8 * kmp, 0.0f);
    for (int o = 0; o < N; o++)
      for (int k = 0; k < K; k++) {
        const int m = k >> 3, e = m & 7;
        wr[(size_t)o * 8 * kmp + (size_t)(k & 7) * kmp + ((m & ~7) | ((e & 1) ? 4 + (e >> 1) : (e >> 1)))] = d->w[l][(size_t)o * K + k];
      }
    float *dwr = nullptr;
    JAMD_HIP(hipMalloc(&dwr, sizeof(float) * wr.size()));
    JAMD_HIP(hipMemcpy(dwr, wr.data(), sizeof(float) * wr.size(), hipMemcpyHostToDevice));
    n->d_wr.push_back(dwr); n->kmp.push_back(kmp);
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
  for (float *p : n->d_wr) (void)hipFree(p);
  if (n->d_xr) (void)hipFree(n->d_xr);
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
  for (int l = 1; l < n->nlayer; l++) if (8 * n->kmp[l] > maxh) maxh = 8 * n->kmp[l];
  if ((rc = ensure(&n->d_xr, &n->xr_cap, sizeof(float) * (size_t)T * 8 * n->kmp[0])) != JAMD_OK) return rc;
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
  int per = ((T + nchunk - 1) / nchunk + RB - 1) / RB * RB;
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
    {
      const int L0 = 8 * n->kmp[0], nmb = (Tc + RB - 1) / RB;
      float *xr = n->d_xr + (size_t)t0 * L0;
      hipLaunchKernelGGL(dnn_pack_rm_kernel, dim3((L0 + 255) / 256, Tc < 16384 ? Tc : 16384), dim3(256), 0, st,
                         dev_frames + (size_t)t0 * n->dims[0], xr, Tc, n->dims[0], n->dims[0], n->kmp[0]);
      const float *src = xr;
      for (int l = 0; l < n->nlayer; l++) {
        const int N = n->dims[l + 1];
        const bool last = (l == n->nlayer - 1);
        const int kout = last ? 0 : n->kmp[l + 1];
        const int ncover = last ? N : (8 * kout > N ? 8 * kout : N);
        float *dst = last ? dev_out + (size_t)t0 * S : n->d_act[l & 1];
        const int nnb = (ncover + RB - 1) / RB;
        const int grid = 8 * ((nnb + 7) / 8) * nmb;
        const bool straddle = (n->kmp[l] % MS) != 0;     // segments that are not whole slabs: stream the row (see the kernel)
#define JAMD_DNN_LAYER(OUT_, STR_, LDY_, KOUT_)                                                                          \
        hipLaunchKernelGGL((dnn_layer_rs_kernel<OUT_, STR_>), dim3(grid), dim3(256), 0, st, src, n->d_wr[l], n->d_b[l], \
                           n->eng->d_logistic, dst, Tc, n->kmp[l], N, ncover, LDY_, KOUT_, nmb, n->d_zero)
        if (last) { if (straddle) JAMD_DNN_LAYER(0, true, N, 0); else JAMD_DNN_LAYER(0, false, N, 0); }
        else { if (straddle) JAMD_DNN_LAYER(1, true, 8 * kout, kout); else JAMD_DNN_LAYER(1, false, 8 * kout, kout); }
#undef JAMD_DNN_LAYER
        src = dst;
      }
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
