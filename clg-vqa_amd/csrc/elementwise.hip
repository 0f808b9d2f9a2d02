// HBM-bound elementwise / layout kernels: sparse-fine-tuning mask products, bf16 (hi, lo) splits, weight
// preparation, bf16 transposes, column sums, the additive attention mask, UC2 embedding gather / scatter, the
// box-location projection, and the fused AdamW step.  All of them stream 16 B (or 8 B for bf16) per lane.
#include "common.h"
#include "../../include/vlhip.h"

namespace {

inline unsigned grid_for(int64_t work_items, int threads, int64_t cap = 256 * 8) {
  int64_t g = (work_items + threads - 1) / threads;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ---- SFT: out = a (*) m  (train_task_sft.py:128-132 via torch prune.py:20-31; autograd = same product) ------
__global__ void mask_mul_kernel(const float* __restrict__ a, const float* __restrict__ m, float* __restrict__ out,
                                long n4, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 x = reinterpret_cast<const float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(m)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
  }
  if (blockIdx.x == 0) {  // tail (n not a multiple of 4)
    const long i = n4 * 4 + threadIdx.x;
    if (i < n) out[i] = a[i] * m[i];
  }
}

// ---- fp32 -> (hi, lo) bf16 split --------------------------------------------------------------------------------
__global__ void split_kernel(const float* __restrict__ x, bf16_raw* __restrict__ hi, bf16_raw* __restrict__ lo,
                             long n4, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    ushort4 h, l;
    split_bf16(v.x, h.x, l.x); split_bf16(v.y, h.y, l.y); split_bf16(v.z, h.z, l.z); split_bf16(v.w, h.w, l.w);
    reinterpret_cast<ushort4*>(hi)[i] = h;
    if (lo) reinterpret_cast<ushort4*>(lo)[i] = l;
  }
  if (blockIdx.x == 0) {
    const long i = n4 * 4 + threadIdx.x;
    if (i < n) {
      bf16_raw h, l;
      split_bf16(x[i], h, l);
      hi[i] = h;
      if (lo) lo[i] = l;
    }
  }
}

// ---- weight prep: W[N,K] (*mask) -> w_hi, w_lo [N,K] (ld = ldw) and wt_hi [K,N] (ld = ldt) -------------------
__device__ __forceinline__ void weight_prep_tile(const float* __restrict__ w, const float* __restrict__ mask,
                                                 bf16_raw* w_hi, bf16_raw* w_lo, bf16_raw* wt_hi, int N, int K,
                                                 long ldw, long ldt, int n0, int k0, bf16_raw (*tile)[66]) {
  // fast path (whole tile inside the matrix, 16-byte-aligned rows): 16-byte loads, 8-byte stores, 4 elements per thread
  // and pass; the scalar path below serves ragged edges (the 1842-label classifier) and odd leading dimensions
  const bool vec = n0 + 64 <= N && k0 + 64 <= K && (K & 3) == 0 && (ldw & 3) == 0 && (ldt & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(mask)) & 15) == 0 &&
                   ((reinterpret_cast<uintptr_t>(w_hi) | reinterpret_cast<uintptr_t>(w_lo) | reinterpret_cast<uintptr_t>(wt_hi)) & 7) == 0;
  if (vec) {
    const int c4 = (threadIdx.x & 15) * 4, r = threadIdx.x >> 4;  // 16 threads per row, 16 rows per pass
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int rr = pass * 16 + r;
      const long src = (long)(n0 + rr) * K + k0 + c4;
      float4 v = *reinterpret_cast<const float4*>(w + src);
      if (mask) {
        const float4 mk = *reinterpret_cast<const float4*>(mask + src);
        v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
      }
      ushort4 h, l;
      split_bf16(v.x, h.x, l.x); split_bf16(v.y, h.y, l.y); split_bf16(v.z, h.z, l.z); split_bf16(v.w, h.w, l.w);
      const long dst = (long)(n0 + rr) * ldw + k0 + c4;
      if (w_hi) *reinterpret_cast<ushort4*>(w_hi + dst) = h;
      if (w_lo) *reinterpret_cast<ushort4*>(w_lo + dst) = l;
      tile[rr][c4] = h.x; tile[rr][c4 + 1] = h.y; tile[rr][c4 + 2] = h.z; tile[rr][c4 + 3] = h.w;
    }
    if (!wt_hi) return;
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int kk = pass * 16 + r;  // row of the transposed tile
      ushort4 t;
      t.x = tile[c4][kk]; t.y = tile[c4 + 1][kk]; t.z = tile[c4 + 2][kk]; t.w = tile[c4 + 3][kk];
      *reinterpret_cast<ushort4*>(wt_hi + (long)(k0 + kk) * ldt + n0 + c4) = t;
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int n = n0 + rr, k = k0 + tx;
    bf16_raw h = 0;
    if (n < N && k < K) {
      float v = w[(long)n * K + k];
      if (mask) v *= mask[(long)n * K + k];
      bf16_raw l;
      split_bf16(v, h, l);
      if (w_hi) w_hi[(long)n * ldw + k] = h;
      if (w_lo) w_lo[(long)n * ldw + k] = l;
    }
    tile[rr][tx] = h;
  }
  if (!wt_hi) return;
  __syncthreads();
  for (int rr = ty; rr < 64; rr += 4) {
    const int k = k0 + rr, n = n0 + tx;
    if (k < K && n < N) wt_hi[(long)k * ldt + n] = tile[tx][rr];
  }
}
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ w, const float* __restrict__ mask,
                                                          bf16_raw* w_hi, bf16_raw* w_lo, bf16_raw* wt_hi, int N,
                                                          int K, long ldw, long ldt) {
  __shared__ bf16_raw tile[64][66];
  weight_prep_tile(w, mask, w_hi, w_lo, wt_hi, N, K, ldw, ldt, blockIdx.y * 64, blockIdx.x * 64, tile);
}
// all Linear weights of the model in ONE launch: a device table of 10 x int64 per weight
// [w32, mask32, w_hi, w_lo, wt_hi, N, K, ldw, ldt, first_tile]; blockIdx.x is a global 64x64-tile index.
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const int64_t* __restrict__ tab, int nd) {
  __shared__ bf16_raw tile[64][66];
  int i = 0;
  while (i + 1 < nd && tab[(i + 1) * 10 + 9] <= (int64_t)blockIdx.x) ++i;
  const int64_t* d = tab + i * 10;
  const int N = (int)d[5], K = (int)d[6];
  const int t = (int)(blockIdx.x - d[9]);
  const int tiles_k = (K + 63) / 64;
  weight_prep_tile(reinterpret_cast<const float*>(d[0]), reinterpret_cast<const float*>(d[1]),
                   reinterpret_cast<bf16_raw*>(d[2]), reinterpret_cast<bf16_raw*>(d[3]),
                   reinterpret_cast<bf16_raw*>(d[4]), N, K, d[7], d[8], (t / tiles_k) * 64, (t % tiles_k) * 64, tile);
}

// ---- bf16 transpose [M,N] -> [N,M] ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_raw* __restrict__ in, bf16_raw* __restrict__ out,
                                                        int M, int N, long ld_in, long ld_out) {
  __shared__ bf16_raw tile[64][66];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int m = m0 + rr, n = n0 + tx;
    tile[rr][tx] = (m < M && n < N) ? in[(long)m * ld_in + n] : (bf16_raw)0;
  }
  __syncthreads();
  for (int rr = ty; rr < 64; rr += 4) {
    const int n = n0 + rr, m = m0 + tx;
    if (n < N && m < M) out[(long)n * ld_out + m] = tile[tx][rr];
  }
}

// ---- column sums of a bf16 matrix (bias gradients) ------------------------------------------------------------
constexpr int CS_ROWBLOCKS = 128;
__global__ __launch_bounds__(256) void colsum_stage1(const bf16_raw* __restrict__ x, int M, int N, long ld,
                                                     float* __restrict__ ws, int vec8) {
  // 256 columns per workgroup: 32 column groups of 8 (one 16-byte load per row) x 8 row lanes
  __shared__ float red[8][256 + 8];
  const int cg = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n = blockIdx.x * 256 + cg * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    if (vec8 && n + 7 < N) {
      for (int m = blockIdx.y * 8 + ty; m < M; m += gridDim.y * 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + (long)m * ld + n);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[2 * t] += __uint_as_float(w[t] << 16);
          acc[2 * t + 1] += __uint_as_float(w[t] & 0xFFFF0000u);
        }
      }
    } else {
      for (int m = blockIdx.y * 8 + ty; m < M; m += gridDim.y * 8)
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (n + t < N) acc[t] += bf16_to_f32(x[(long)m * ld + n + t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) red[ty][cg * 8 + t] = acc[t];
  __syncthreads();
  const int c = threadIdx.x;  // one column per thread
  if (blockIdx.x * 256 + c < N) {
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s2 += red[t][c];
    ws[(long)blockIdx.y * N + blockIdx.x * 256 + c] = s2;
  }
}
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ ws, int nrb, int N,
                                                     float* __restrict__ out) {
  __shared__ float red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + tx;
  float s = 0.f;
  if (n < N)
    for (int b = ty; b < nrb; b += 16) s += ws[(long)b * N + n];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][tx];
    out[n] = t;
  }
}

// ---- additive key mask over [text ; boxes] (encoders.py:978-995) --------------------------------------------
__global__ void addmask_kernel(const int64_t* __restrict__ tm, const int64_t* __restrict__ im, float* __restrict__ out,
                               int B, int T, int V) {
  const int S = T + V;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * S) return;
  const int b = (int)(i / S), s = (int)(i - (long)b * S);
  const float m = s < T ? (float)tm[(long)b * T + s] : (float)im[(long)b * V + (s - T)];
  out[i] = (1.0f - m) * -10000.0f;
}

// ---- UC2 text embedding gather (embeddings.py:648-653) ---------------------------------------------------------
__device__ __forceinline__ int roberta_pos(const int64_t* ids_row, int t, int T, int64_t pad, int lane) {
  int cnt = 0;  // number of non-pad ids in [0, t]
  for (int base = 0; base <= t; base += 64) {
    const int i = base + lane;
    const bool nz = (i <= t) && (ids_row[i] != pad);
    cnt += __popcll(__ballot(nz));
  }
  return (ids_row[t] != pad ? cnt : 0) + (int)pad;
}
__global__ __launch_bounds__(256) void embed_text_fwd_kernel(const int64_t* __restrict__ ids,
                                                             const int64_t* __restrict__ seg,
                                                             const float* __restrict__ word,
                                                             const float* __restrict__ pos,
                                                             const float* __restrict__ type, float* __restrict__ z,
                                                             int B, int T, int H, int64_t pad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= (long)B * T) return;
  const int b = (int)(r / T), t = (int)(r - (long)b * T);
  const int pid = roberta_pos(ids + (long)b * T, t, T, pad, lane);
  const float* wr = word + ids[r] * (long)H;
  const float* pr = pos + (long)pid * H;
  const float* tr = type + seg[r] * (long)H;
  for (int c = lane * 4; c < H; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(wr + c);
    const float4 d = *reinterpret_cast<const float4*>(pr + c);
    const float4 e = *reinterpret_cast<const float4*>(tr + c);
    *reinterpret_cast<float4*>(z + r * H + c) =
        make_float4(a.x + d.x + e.x, a.y + d.y + e.y, a.z + d.z + e.z, a.w + d.w + e.w);
  }
}
// scatter-add into the dense tables (nn.Embedding(sparse=False), embeddings.py:617); the pad row of
// word_embeddings receives no gradient (padding_idx).  One wave per token: 256 contiguous bytes per atomic
// wave-instruction (MI355X float-atomic sweet spot).
// One wave per (position t, 16 samples): consecutive samples that hit the same table row (the same position id at a
// given t, token type 0, `<s>` at t = 0, the pad position) are summed in registers and flushed with ONE atomic per
// column when the row changes, instead of one atomic per (sample, column) -- the position / type rows saw 256-way
// and 5120-way same-address contention (220 us per step); sums are re-associated, nothing else changes.
constexpr int ETB_SAMPLES = 16, ETB_MAXV = 32;  // H <= 64 * ETB_MAXV
__global__ __launch_bounds__(256) void embed_text_bwd_kernel(const int64_t* __restrict__ ids,
                                                             const int64_t* __restrict__ seg,
                                                             const float* __restrict__ dz, float* dword, float* dpos,
                                                             float* dtype, int B, int T, int H, int64_t pad,
                                                             unsigned char* row_flags) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups = (B + ETB_SAMPLES - 1) / ETB_SAMPLES;
  const long gw = (long)blockIdx.x * 4 + wave;  // (t, sample group)
  if (gw >= (long)T * groups) return;
  const int t = (int)(gw / groups), b0 = (int)(gw - (long)t * groups) * ETB_SAMPLES;
  const int nv = H / 64;
  float aw[ETB_MAXV], ap[ETB_MAXV], at[ETB_MAXV];
  long cw = -1, cp = -1, ct = -1;  // table rows the three accumulators belong to
  auto flush = [&](float* table, long row, float* acc) {
    if (row >= 0 && table) {
      float* dst = table + row * (long)H;
#pragma unroll
      for (int j = 0; j < ETB_MAXV; ++j)
        if (j < nv) atomicAdd(dst + lane + 64 * j, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < ETB_MAXV; ++j) acc[j] = 0.f;
  };
#pragma unroll
  for (int j = 0; j < ETB_MAXV; ++j) aw[j] = ap[j] = at[j] = 0.f;
  for (int b = b0; b < min(B, b0 + ETB_SAMPLES); ++b) {
    const long r = (long)b * T + t;
    const int pid = roberta_pos(ids + (long)b * T, t, T, pad, lane);
    const int64_t id = ids[r];
    const long wrow = (dword && id != pad) ? (long)id : -1;  // the pad row receives no gradient
    const long trow = (long)seg[r];
    if (wrow != cw) { flush(dword, cw, aw); cw = wrow; }
    if (pid != cp) { flush(dpos, cp, ap); cp = pid; }
    if (trow != ct) { flush(dtype, ct, at); ct = trow; }
    if (row_flags && wrow >= 0 && lane == 0) row_flags[id] = 1;  // this table row now carries optimizer state
#pragma unroll
    for (int j = 0; j < ETB_MAXV; ++j)
      if (j < nv) {
        const float g = dz[r * H + lane + 64 * j];
        aw[j] += g; ap[j] += g; at[j] += g;
      }
  }
  flush(dword, cw, aw);
  flush(dpos, cp, ap);
  flush(dtype, ct, at);
}

// ---- plain row gather / scatter-add (M3P text embedding: tensor = embeddings(x), m3p_transformer.py:908) -------
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tab,
                                                          float* __restrict__ out, long R, int H) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const float* src = tab + ids[r] * (long)H;
  for (int c = lane * 4; c < H; c += 256)
    *reinterpret_cast<float4*>(out + r * H + c) = *reinterpret_cast<const float4*>(src + c);
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dz,
                                                           float* dtab, long R, int H, int64_t pad,
                                                           unsigned char* row_flags) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int64_t id = ids[r];
  if (id == pad || id < 0) return;  // nn.Embedding(padding_idx): the pad row receives no gradient; -1 = empty slot
  if (row_flags && lane == 0) row_flags[id] = 1;
  float* dst = dtab + id * (long)H;
  for (int c = lane; c < H; c += 64) atomicAdd(dst + c, dz[r * H + c]);
}

// ---- box-location projection (embeddings.py:661: Linear(num_locs -> H)) ---------------------------------------
// a thread owns 4 consecutive columns of 4 rows: the 4 x L weights of its columns are read once, every row costs L
// broadcast loads and one 16-byte store (one thread per element with an integer division and a 28-byte-strided weight
// walk ran at 0.7 TB/s: 42 us for the 28 MB of c2)
__global__ __launch_bounds__(256) void loc_fwd_kernel(const float* __restrict__ loc, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ y, long R, int L, int H) {
  const int cq = H >> 2;                       // column quads per row (H % 4 == 0, checked by the host)
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long rg = t / cq;                      // group of 4 rows
  const int c = (int)(t - rg * cq) * 4;
  if (rg * 4 >= R) return;
  float wr[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int l = 0; l < 8; ++l) wr[j][l] = l < L ? w[(c + j) * L + l] : 0.f;
  const float4 bb = *reinterpret_cast<const float4*>(b + c);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long r = rg * 4 + i;
    if (r >= R) break;
    float x[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) x[l] = l < L ? loc[r * L + l] : 0.f;
    float o[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int l = 0; l < 8; ++l) o[j] += x[l] * wr[j][l];  // (same order of additions as the scalar form: l ascending)
    *reinterpret_cast<float4*>(y + r * H + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
// dW[c][l] += sum_r dy[r][c] loc[r][l], db[c] += sum_r dy[r][c].  Workgroup = 64 rows x 64 columns: thread (tx = column, ty =
// one of 4 row groups) walks 16 rows, the 4 groups are combined in LDS and 64 threads issue the atomics -- 12 x more
// workgroups and 4 x shorter dependent chains than one workgroup per 64 full rows (145 us at c2), the same number of atomics.
// ws != NULL: the deterministic form -- every workgroup stores its partial sums to ws[row block][l][c] (l = 8: the bias) and
// loc_bwd_reduce_kernel adds the row blocks in order; ws == NULL: float atomics straight into dw / db.
__global__ __launch_bounds__(256) void loc_bwd_kernel(const float* __restrict__ loc, const float* __restrict__ dy,
                                                      float* dw, float* db, long R, int L, int H, float* ws) {
  __shared__ float red[4][9][64];
  // (the row group is the wave index: made scalar, so that the 7 box coordinates of a row -- the same for all 64 lanes -- come in
  // through scalar loads instead of 7 vector loads per lane and row)
  const int tx = threadIdx.x & 63, ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cblocks = (H + 63) / 64;
  const long rb = blockIdx.x / cblocks;
  const int c = (int)(blockIdx.x - rb * cblocks) * 64 + tx;
  const long r0 = rb * 64 + ty * 16;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float accb = 0.f;
  if (c < H) {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const long r = r0 + i;
      if (r >= R) break;
      const float g = dy[r * H + c];
      accb += g;
#pragma unroll
      for (int l = 0; l < 8; ++l)
        if (l < L) acc[l] += g * loc[r * L + l];
    }
  }
#pragma unroll
  for (int l = 0; l < 8; ++l) red[ty][l][tx] = acc[l];
  red[ty][8][tx] = accb;
  __syncthreads();
  if (ty == 0 && c < H) {
    if (ws) {
      float* dst = ws + rb * 9 * (long)H + c;
      dst[8 * (long)H] = (red[0][8][tx] + red[1][8][tx]) + (red[2][8][tx] + red[3][8][tx]);
#pragma unroll
      for (int l = 0; l < 8; ++l)
        if (l < L) dst[l * (long)H] = (red[0][l][tx] + red[1][l][tx]) + (red[2][l][tx] + red[3][l][tx]);
      return;
    }
    atomicAdd(db + c, (red[0][8][tx] + red[1][8][tx]) + (red[2][8][tx] + red[3][8][tx]));
#pragma unroll
    for (int l = 0; l < 8; ++l)
      if (l < L) atomicAdd(dw + c * L + l, (red[0][l][tx] + red[1][l][tx]) + (red[2][l][tx] + red[3][l][tx]));
  }
}
// dw[c][l] += sum over the row blocks, db likewise from slot l = 8.  Fixed order: wave w adds the blocks b = w (mod 4) in
// ascending order (two independent chains), the four partial sums are combined ((0 + 1) + (2 + 3)) through LDS -- a quarter of
// the dependent chain of one thread per output (23 -> 8 us at c2: the launch sits on the critical path at the end of backward).
__global__ __launch_bounds__(256) void loc_bwd_reduce_kernel(const float* __restrict__ ws, float* dw, float* db, int nrb, int L,
                                                             int H) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx, l = blockIdx.y;  // l in [0, L] (L = the bias slot 8)
  const int slot = l < L ? l : 8;
  float s0 = 0.f, s1 = 0.f;
  if (c < H) {
    const float* src = ws + slot * (long)H + c;
    int b = w;
    for (; b + 4 < nrb; b += 8) {
      s0 += src[(long)b * 9 * H];
      s1 += src[(long)(b + 4) * 9 * H];
    }
    if (b < nrb) s0 += src[(long)b * 9 * H];
  }
  red[w][tx] = s0 + s1;
  __syncthreads();
  if (w == 0 && c < H) {
    const float t = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    if (l < L) dw[c * L + l] += t;
    else db[c] += t;
  }
}

// ---- fused AdamW over a flat arena ---------------------------------------------------------------------------
// pytorch_transformers.optimization.AdamW semantics (un-vendored dependency of the reference; call site
// train_task.py:264-268): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= step * m / (sqrt(v) + eps) with
// step = lr (* sqrt(1-b2^t)/(1-b1^t) when correct_bias); then decoupled decay p -= lr * wd * p.
constexpr int ADAMW_MAX_SEG = 2048;
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n4,
                                                    const int64_t* __restrict__ seg_end,
                                                    const float* __restrict__ seg_lr,
                                                    const float* __restrict__ seg_wd, int nseg, float b1, float b2,
                                                    float eps, float bc, float lr_mult, const float* gscale_ptr,
                                                    float gscale_const, const float* sumsq_ptr, float max_norm, float post,
                                                    float* sumsq_next, int zero_grad,
                                                    const unsigned char* __restrict__ row_flags, long fl_beg4, long fl_end4,
                                                    int fl_row4, const int64_t* __restrict__ seg_step, int correct_bias) {
  // segment table (every segment starts on a multiple of 4 elements) -> LDS, in float4 units
  __shared__ int s_end4[ADAMW_MAX_SEG];
  __shared__ float s_lr[ADAMW_MAX_SEG], s_wd[ADAMW_MAX_SEG], s_bc[ADAMW_MAX_SEG];
  for (int i = threadIdx.x; i < nseg; i += 256) {
    s_end4[i] = (int)(seg_end[i] >> 2);
    s_lr[i] = seg_lr[i] < 0.f ? -1.f : seg_lr[i] * lr_mult;
    s_wd[i] = seg_wd[i];
    // per-segment step counts (pytorch_transformers.AdamW keeps state['step'] per parameter and advances it only when
    // that parameter has a gradient): the same double-precision expression the host evaluates for the uniform case
    float b = bc;
    if (seg_step) {
      const double t = (double)(seg_step[i] > 0 ? seg_step[i] : 1);
      b = correct_bias ? (float)(sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t))) : 1.0f;
    }
    s_bc[i] = b;
  }
  __syncthreads();
  float gs = gscale_ptr ? *gscale_ptr : gscale_const;
  if (sumsq_ptr) {
    // clip_grad_norm_ (train_task.py:330) on the device: ||g|| of the averaged gradient = sqrt(sum g^2) * post;
    // coefficient min(1, max_norm / (||g|| + 1e-6)), times the 1/world factor `post` of the all-reduce
    const float norm = sqrtf(*sumsq_ptr) * post;
    gs = fminf(max_norm / (norm + 1e-6f), 1.0f) * post;
  }
  if (sumsq_next && blockIdx.x == 0 && threadIdx.x == 0) *sumsq_next = 0.f;  // the NEXT step's accumulator (nobody reads it now)
  const float c1 = 1.f - b1, c2 = 1.f - b2;
  const long stride = (long)gridDim.x * blockDim.x;
  auto seg_of = [&](long i) {  // first segment whose end > i
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_end4[mid] > i) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  auto update4 = [&](long i, float lr, float wd, float bcs) {
    const float step = lr * bcs, decay = 1.f - lr * wd;
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
#define VL_ADAM1(c)                                          \
    {                                                        \
      const float grad = gg.c * gs;                          \
      mm.c = b1 * mm.c + c1 * grad;                          \
      vv.c = b2 * vv.c + c2 * grad * grad;                   \
      float x = pp.c - step * (mm.c / (sqrtf(vv.c) + eps)); \
      if (wd > 0.f) x *= decay;                              \
      pp.c = x;                                              \
    }
    VL_ADAM1(x) VL_ADAM1(y) VL_ADAM1(z) VL_ADAM1(w)
#undef VL_ADAM1
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  // dense part: everything outside the flagged (embedding-table) range
  const long fl_len4 = row_flags ? fl_end4 - fl_beg4 : 0;
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n4 - fl_len4; j += stride) {
    const long i = (row_flags && j >= fl_beg4) ? j + fl_len4 : j;
    const int sg = seg_of(i);
    // a NEGATIVE segment learning rate marks a parameter that received no gradient: skipped entirely (no moment decay, no
    // weight decay, no traffic), like `if p.grad is None: continue` in pytorch_transformers.AdamW -- M3P's 93 M never-used
    // parameters are 2.6 GB per step otherwise
    if (s_lr[sg] < 0.f) continue;
    update4(i, s_lr[sg], s_wd[sg], s_bc[sg]);
  }
  // flagged table, one wave per row: rows that never received a gradient have g = m = v = 0, so their AdamW update is
  // exactly p *= decay (8 B/param instead of 32 B/param, bit-identical to the dense path) -- and with the reference's
  // hyper-parameters lr * wd = 4e-5 * 1e-4 = 4e-9 is below half an fp32 ulp of 1: `decay` IS 1.0f, the multiply (like
  // the reference's p.add_(-lr * wd, p)) changes no bit, and such rows cost one flag byte
  if (row_flags) {
    const int sg = seg_of(fl_beg4);
    const float lr = s_lr[sg], wd = s_wd[sg], decay = 1.f - lr * wd;
    if (lr < 0.f) return;  // (the table itself received no gradient at all)
    const bool noop = !(wd > 0.f && lr != 0.f && decay != 1.0f);
    const int lane = threadIdx.x & 63;
    const long nrows = fl_len4 / fl_row4;
    const long nwaves = stride >> 6;
    for (long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < nrows; row += nwaves) {
      const long base = fl_beg4 + row * fl_row4;
      if (row_flags[row]) {
        for (int c = lane; c < fl_row4; c += 64) update4(base + c, lr, wd, s_bc[sg]);
      } else if (!noop) {
        for (int c = lane; c < fl_row4; c += 64) {
          float4 pp = reinterpret_cast<float4*>(p)[base + c];
          pp.x *= decay; pp.y *= decay; pp.z *= decay; pp.w *= decay;
          reinterpret_cast<float4*>(p)[base + c] = pp;
        }
      }
    }
  }
}

// out += sum(x^2).  Optional flagged range (the word-embedding gradient inside the flat arena): rows whose flag is 0
// never received a gradient, their gradient is exactly zero and they are not read (0.77 GB of zeros per step at c2).
// part != NULL: the deterministic form -- block b stores its sum to part[b] and sumsq_finish_kernel adds the blocks in order.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* out,
                                                    const unsigned char* __restrict__ row_flags, long fl_beg4,
                                                    long fl_end4, int fl_row4, float* part) {
  __shared__ float red[4];
  float s = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = n >> 2;
  const long fl_len4 = row_flags ? fl_end4 - fl_beg4 : 0;
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n4 - fl_len4; j += stride) {
    const long i = (row_flags && j >= fl_beg4) ? j + fl_len4 : j;
    const float4 q = reinterpret_cast<const float4*>(x)[i];
    s += (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
  }
  if (row_flags) {
    const int lane = threadIdx.x & 63;
    const long nrows = fl_len4 / fl_row4, nwaves = stride >> 6;
    for (long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < nrows; row += nwaves) {
      if (!row_flags[row]) continue;
      const float4* r = reinterpret_cast<const float4*>(x) + fl_beg4 + row * fl_row4;
      for (int c = lane; c < fl_row4; c += 64) {
        const float4 q = r[c];
        s += (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
      }
    }
  }
  if (blockIdx.x == 0) {
    const long i = n4 * 4 + threadIdx.x;
    if (i < n) s += x[i] * x[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    if (part) part[blockIdx.x] = t;
    else atomicAdd(out, t);
  }
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ part, int nblocks, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += part[i];  // fixed assignment, fixed order
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out += (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" int vl_mask_mul(const float* a, const float* m, float* out, int64_t n, void* stream) {
  VL_CHECK_ARG(a && m && out && n >= 0, "vl_mask_mul: bad arguments");
  if (n == 0) return 0;
  VL_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)m & 15) == 0 && ((uintptr_t)out & 15) == 0,
               "vl_mask_mul: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(mask_mul_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, a, m, out,
                     (long)(n / 4), (long)n);
  VL_CHECK_LAUNCH("vl_mask_mul");
  return 0;
}

extern "C" int vl_split_f32(const float* x32, void* hi, void* lo, int64_t n, void* stream) {
  VL_CHECK_ARG(x32 && hi && n >= 0, "vl_split_f32: bad arguments");
  if (n == 0) return 0;
  VL_CHECK_ARG(((uintptr_t)x32 & 15) == 0 && ((uintptr_t)hi & 7) == 0 && ((uintptr_t)lo & 7) == 0,
               "vl_split_f32: misaligned pointers");
  hipLaunchKernelGGL(split_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x32,
                     (bf16_raw*)hi, (bf16_raw*)lo, (long)(n / 4), (long)n);
  VL_CHECK_LAUNCH("vl_split_f32");
  return 0;
}

extern "C" int vl_weight_prep(const float* w32, const float* mask32, void* w_hi, void* w_lo, void* wt_hi, int64_t N,
                              int64_t K, int64_t ldw, int64_t ldt, void* stream) {
  VL_CHECK_ARG(w32 && N > 0 && K > 0 && (w_hi || w_lo || wt_hi), "vl_weight_prep: bad arguments");
  VL_CHECK_ARG(ldw >= K && (!wt_hi || ldt >= N), "vl_weight_prep: leading dimensions too small");
  hipLaunchKernelGGL(weight_prep_kernel, dim3((unsigned)((K + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, w32, mask32, (bf16_raw*)w_hi, (bf16_raw*)w_lo, (bf16_raw*)wt_hi, (int)N,
                     (int)K, (long)ldw, (long)ldt);
  VL_CHECK_LAUNCH("vl_weight_prep");
  return 0;
}

extern "C" int vl_weight_prep_multi(const int64_t* table_dev, int64_t ndesc, int64_t total_tiles, void* stream) {
  VL_CHECK_ARG(table_dev && ndesc > 0 && total_tiles > 0 && total_tiles < (1LL << 31), "vl_weight_prep_multi: bad arguments");
  hipLaunchKernelGGL(weight_prep_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, table_dev,
                     (int)ndesc);
  VL_CHECK_LAUNCH("vl_weight_prep_multi");
  return 0;
}

extern "C" int vl_transpose_bf16(const void* in, void* out, int64_t M, int64_t N, int64_t ld_in, int64_t ld_out,
                                 void* stream) {
  VL_CHECK_ARG(in && out && M > 0 && N > 0 && ld_in >= N && ld_out >= M, "vl_transpose_bf16: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_raw*)in, (bf16_raw*)out, (int)M, (int)N, (long)ld_in,
                     (long)ld_out);
  VL_CHECK_LAUNCH("vl_transpose_bf16");
  return 0;
}

extern "C" int64_t vl_colsum_ws_floats(int64_t M, int64_t N) {
  int64_t rb = (M + 7) / 8;
  if (rb > CS_ROWBLOCKS) rb = CS_ROWBLOCKS;
  return rb * N;
}
extern "C" int vl_colsum_bf16(const void* x16, int64_t M, int64_t N, int64_t ld, float* ws, float* out32,
                              void* stream) {
  VL_CHECK_ARG(x16 && ws && out32 && M > 0 && N > 0, "vl_colsum_bf16: bad arguments");
  VL_CHECK_ARG(N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x16 & 7) == 0, "vl_colsum_bf16: N, ld must be multiples of 4");
  int64_t rb = (M + 7) / 8;
  if (rb > CS_ROWBLOCKS) rb = CS_ROWBLOCKS;
  const int vec8 = (ld % 8 == 0) && (((uintptr_t)x16 & 15) == 0);
  hipLaunchKernelGGL(colsum_stage1, dim3((unsigned)((N + 255) / 256), (unsigned)rb), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_raw*)x16, (int)M, (int)N, (long)ld, ws, vec8);
  VL_CHECK_LAUNCH("vl_colsum_bf16");
  hipLaunchKernelGGL(colsum_stage2, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, (hipStream_t)stream, ws, (int)rb,
                     (int)N, out32);
  VL_CHECK_LAUNCH("vl_colsum_bf16(stage2)");
  return 0;
}

extern "C" int vl_addmask(const int64_t* text_mask, const int64_t* img_mask, float* addmask, int64_t B, int64_t T,
                          int64_t V, void* stream) {
  VL_CHECK_ARG(text_mask && img_mask && addmask && B > 0 && T > 0 && V > 0, "vl_addmask: bad arguments");
  const int64_t n = B * (T + V);
  hipLaunchKernelGGL(addmask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, text_mask,
                     img_mask, addmask, (int)B, (int)T, (int)V);
  VL_CHECK_LAUNCH("vl_addmask");
  return 0;
}

extern "C" int vl_embed_text_fwd(const int64_t* ids, const int64_t* seg, const float* word, const float* pos,
                                 const float* type, float* z32, int64_t B, int64_t T, int64_t H, int64_t pad_id,
                                 void* stream) {
  VL_CHECK_ARG(ids && seg && word && pos && type && z32 && B > 0 && T > 0 && H > 0 && H % 4 == 0,
               "vl_embed_text_fwd: bad arguments (H must be a multiple of 4)");
  hipLaunchKernelGGL(embed_text_fwd_kernel, dim3((unsigned)((B * T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ids,
                     seg, word, pos, type, z32, (int)B, (int)T, (int)H, pad_id);
  VL_CHECK_LAUNCH("vl_embed_text_fwd");
  return 0;
}
extern "C" int vl_embed_text_bwd(const int64_t* ids, const int64_t* seg, const float* dz32, float* dword, float* dpos,
                                 float* dtype, int64_t B, int64_t T, int64_t H, int64_t pad_id, uint8_t* row_flags,
                                 void* stream) {
  VL_CHECK_ARG(ids && seg && dz32 && dpos && dtype && B > 0 && T > 0 && H > 0, "vl_embed_text_bwd: bad arguments");
  VL_CHECK_ARG(H % 64 == 0 && H <= 64 * ETB_MAXV, "vl_embed_text_bwd: H must be a multiple of 64, at most %d", 64 * ETB_MAXV);
  const int64_t waves = T * ((B + ETB_SAMPLES - 1) / ETB_SAMPLES);
  hipLaunchKernelGGL(embed_text_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ids,
                     seg, dz32, dword, dpos, dtype, (int)B, (int)T, (int)H, pad_id, row_flags);
  VL_CHECK_LAUNCH("vl_embed_text_bwd");
  return 0;
}

extern "C" int vl_embed_gather_fwd(const int64_t* ids, const float* table, float* out32, int64_t R, int64_t H,
                                   void* stream) {
  VL_CHECK_ARG(ids && table && out32 && R > 0 && H > 0 && H % 4 == 0, "vl_embed_gather_fwd: bad arguments");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ids, table,
                     out32, (long)R, (int)H);
  VL_CHECK_LAUNCH("vl_embed_gather_fwd");
  return 0;
}
extern "C" int vl_embed_scatter_add(const int64_t* ids, const float* dz32, float* dtable, int64_t R, int64_t H,
                                    int64_t pad_id, uint8_t* row_flags, void* stream) {
  VL_CHECK_ARG(ids && dz32 && dtable && R > 0 && H > 0, "vl_embed_scatter_add: bad arguments");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ids, dz32,
                     dtable, (long)R, (int)H, pad_id, row_flags);
  VL_CHECK_LAUNCH("vl_embed_scatter_add");
  return 0;
}

extern "C" int vl_loc_linear_fwd(const float* loc, const float* w, const float* b, float* y32, int64_t R, int64_t L,
                                 int64_t H, void* stream) {
  VL_CHECK_ARG(loc && w && b && y32 && R > 0 && L > 0 && L <= 8 && H > 0 && (H & 3) == 0,
               "vl_loc_linear_fwd: bad arguments (L <= 8, H a multiple of 4)");
  const int64_t threads = ((R + 3) / 4) * (H / 4);
  hipLaunchKernelGGL(loc_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, loc, w,
                     b, y32, (long)R, (int)L, (int)H);
  VL_CHECK_LAUNCH("vl_loc_linear_fwd");
  return 0;
}
extern "C" int64_t vl_loc_bwd_ws_floats(int64_t R, int64_t H) { return R > 0 && H > 0 ? ((R + 63) / 64) * 9 * H : 0; }
extern "C" int vl_loc_linear_bwd(const float* loc, const float* dy32, float* dw, float* db, int64_t R, int64_t L,
                                 int64_t H, float* ws, void* stream) {
  VL_CHECK_ARG(loc && dy32 && dw && db && R > 0 && L > 0 && L <= 8 && H > 0, "vl_loc_linear_bwd: bad arguments (L <= 8)");
  const int64_t nrb = (R + 63) / 64;
  hipLaunchKernelGGL(loc_bwd_kernel, dim3((unsigned)(nrb * ((H + 63) / 64))), dim3(256), 0, (hipStream_t)stream, loc, dy32, dw,
                     db, (long)R, (int)L, (int)H, ws);
  if (ws)
    hipLaunchKernelGGL(loc_bwd_reduce_kernel, dim3((unsigned)((H + 63) / 64), (unsigned)(L + 1)), dim3(256), 0, (hipStream_t)stream,
                       ws, dw, db, (int)nrb, (int)L, (int)H);
  VL_CHECK_LAUNCH("vl_loc_linear_bwd");
  return 0;
}

extern "C" int vl_adamw(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const int64_t* seg_end, const float* seg_lr, const float* seg_wd, int64_t nseg, float beta1,
                        float beta2, float eps, int64_t step, const int64_t* seg_step, int correct_bias, float lr_mult,
                        const float* grad_scale_dev, float grad_scale, const float* sumsq_dev, float max_norm, float post,
                        float* sumsq_next, int zero_grad, const uint8_t* row_flags,
                        int64_t flag_begin, int64_t flag_rows, int64_t flag_row_len, void* stream) {
  VL_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && seg_end && seg_lr && seg_wd && n > 0 && nseg > 0 && step > 0,
               "vl_adamw: bad arguments");
  VL_CHECK_ARG(nseg <= ADAMW_MAX_SEG && (n & 3) == 0 && (n >> 2) < (1LL << 31),
               "vl_adamw: arena length must be a multiple of 4 (segments 4-aligned), nseg <= %d", ADAMW_MAX_SEG);
  VL_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
               "vl_adamw: arenas must be 16-byte aligned");
  VL_CHECK_ARG(!row_flags || (flag_begin % 4 == 0 && flag_row_len % 4 == 0 && flag_row_len > 0 && flag_rows > 0 &&
                              flag_begin + flag_rows * flag_row_len <= n),
               "vl_adamw: flagged segment must be 4-aligned and inside the arena");
  float bc = 1.0f;
  if (correct_bias) bc = (float)(sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step)));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, (long)(n / 4), seg_end, seg_lr, seg_wd, (int)nseg, beta1, beta2, eps, bc, lr_mult,
                     grad_scale_dev, grad_scale, sumsq_dev, max_norm, post, sumsq_next, zero_grad, row_flags, (long)(flag_begin / 4),
                     (long)((flag_begin + flag_rows * flag_row_len) / 4), (int)(flag_row_len / 4), seg_step, correct_bias);
  VL_CHECK_LAUNCH("vl_adamw");
  return 0;
}

extern "C" int64_t vl_sumsq_ws_floats(void) { return 2048; }
static int sumsq_launch(const char* fn, const float* x, int64_t n, float* out, const uint8_t* row_flags, int64_t flag_begin,
                        int64_t flag_rows, int64_t flag_row_len, float* ws, void* stream) {
  VL_CHECK_ARG(!ws || ((uintptr_t)ws & 3) == 0, "%s: bad workspace", fn);
  const unsigned grid = grid_for(n / 4 + 1, 256, 2048);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (long)n, out, (const unsigned char*)row_flags,
                     (long)(flag_begin / 4), (long)((flag_begin + flag_rows * flag_row_len) / 4),
                     (int)(row_flags ? flag_row_len / 4 : 1), ws);
  if (ws) hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, (int)grid, out);
  VL_CHECK_LAUNCH(fn);
  return 0;
}
extern "C" int vl_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream) {
  VL_CHECK_ARG(x && out && n > 0 && ((uintptr_t)x & 15) == 0, "vl_sumsq: bad arguments (x must be 16-byte aligned)");
  return sumsq_launch("vl_sumsq", x, n, out, nullptr, 0, 0, 1, ws, stream);
}
// ... skipping the rows of [flag_begin, flag_begin + flag_rows * flag_row_len) whose flag is 0 (same flagged-table
// arguments as vl_adamw: rows that never received a gradient hold exact zeros)
extern "C" int vl_sumsq_flagged(const float* x, int64_t n, float* out, const uint8_t* row_flags, int64_t flag_begin,
                                int64_t flag_rows, int64_t flag_row_len, float* ws, void* stream) {
  VL_CHECK_ARG(x && out && n > 0 && ((uintptr_t)x & 15) == 0 && row_flags, "vl_sumsq_flagged: bad arguments");
  VL_CHECK_ARG(flag_begin % 4 == 0 && flag_row_len % 4 == 0 && flag_row_len > 0 && flag_rows > 0 &&
               flag_begin + flag_rows * flag_row_len <= n, "vl_sumsq_flagged: flagged range must be 4-aligned and inside x");
  return sumsq_launch("vl_sumsq_flagged", x, n, out, row_flags, flag_begin, flag_rows, flag_row_len, ws, stream);
}
