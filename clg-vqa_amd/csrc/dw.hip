// Weight-gradient path: dW = dY^T X for the six Linear weights of a transformer layer (autograd of nn.Linear,
// volta/volta/encoders.py:229-246 Q/K/V, :411-414 out-proj, :496-501 FFN1, :553-556 FFN2), with the SFT mask product
// grad(weight_orig) = grad(weight) * weight_mask (train_task_sft.py:128-132; torch prune.py:20-31) and gradient
// accumulation in the epilogue.
//
// The reduction dimension of these products is the B*S batch rows (14 336 at c2), which is the SLOW dimension of the
// row-major activations.  Instead of gathering k-strided fragments (the TN kernel of gemm.hip), the operands are first
// re-laid "K-major in blocks of 64 rows":
//
//      XT[mb][n][mi] = X[64*mb + mi][n]        (bf16; rows past M are ZERO, so K-tiles are always full)
//
// by vl_transpose_blocked (one launch for all eight operands of a layer, HBM-bound, on the weight-gradient stream; it
// also leaves the per-block column sums of dY = the bias gradients' partial sums).  In that image the 64 k-values of an
// output row are one contiguous 128-byte line, i.e. exactly the LDS row of the 8-wave ping-pong NT kernel (gemm.hip):
// dW becomes an NT product with a K loop of B*S/64 tiles, accumulated in registers over the WHOLE K range -- no split-K
// slabs, no reduce pass -- and written once, straight into the optimizer's gradient arena.  vl_dw_grouped runs all the
// products of a layer as ONE launch (108 tiles of 256 x 256 at H = 768, I = 3072): it deliberately fills only part of
// the chip, the rest stays with the backward critical path on the main stream.
#include <type_traits>

#include "common.h"
#include "../../include/vlhip.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int MAXT = 8;

// ---------------------------------------------------------------------------------------------------------------
// blocked transpose (+ column-sum partials)
// ---------------------------------------------------------------------------------------------------------------
struct TrEntry { const bf16_raw* src; long ld; int N; bf16_raw* dst; float* colsum; int unit0; };
struct TrArgs { TrEntry e[MAXT]; int n; int M; int mblocks; int total_units; };

constexpr int TP = 144;  // LDS row pitch (bytes) of the 64 x 64 tile: conflict-free transposing reads (attention2.hip)

__global__ __launch_bounds__(256) void transpose_blocked_kernel(TrArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[64 * TP];
  __shared__ float cs[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3, c16 = lane & 15;
  for (int u = blockIdx.x; u < a.total_units; u += gridDim.x) {
    int ei = 0;
#pragma unroll
    for (int i = 1; i < MAXT; ++i)
      if (i < a.n && u >= a.e[i].unit0) ei = i;
    const TrEntry& e = a.e[ei];
    const int local = u - e.unit0;
    const int nblocks = e.N >> 6;
    const int mb = local / nblocks, nb = local - mb * nblocks;
    // load: thread -> rows (tid >> 3) and (tid >> 3) + 32, 16-byte chunk tid & 7
    const int r0 = tid >> 3, ch = tid & 7;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r0 + 32 * i;
      const long grow = (long)mb * 64 + r;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (grow < a.M) v = *reinterpret_cast<const uint4*>(e.src + grow * e.ld + nb * 64 + ch * 8);
      *reinterpret_cast<uint4*>(tile + r * TP + ch * 16) = v;
      if (e.colsum) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[2 * t] += __uint_as_float(w[t] << 16);
          s[2 * t + 1] += __uint_as_float(w[t] & 0xFFFF0000u);
        }
      }
    }
    if (e.colsum) {  // rows of one chunk live in lanes with equal (lane & 7): fixed-order tree, then the 4 waves in order
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        s[t] += __shfl_xor(s[t], 8, 64);
        s[t] += __shfl_xor(s[t], 16, 64);
        s[t] += __shfl_xor(s[t], 32, 64);
      }
      if (lane < 8) {
#pragma unroll
        for (int t = 0; t < 8; ++t) cs[wave][lane * 8 + t] = s[t];
      }
    }
    __syncthreads();
    if (e.colsum && tid < 64)
      e.colsum[(long)mb * e.N + nb * 64 + tid] = (cs[0][tid] + cs[1][tid]) + (cs[2][tid] + cs[3][tid]);
    // transposing reads: wave w takes columns [16 w, 16 w + 16); lane (c16, g) receives rows 8 g .. 8 g + 7 (+ 32 half)
    bf16_raw* dst = e.dst + ((long)mb * e.N + nb * 64 + wave * 16 + c16) * 64;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const unsigned char* p0 = tile + (32 * half + 8 * g + qq) * TP + (16 * wave + 4 * pp) * 2;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * TP));
      const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      *reinterpret_cast<s16x8*>(dst + 32 * half + 8 * g) = v;
    }
    __syncthreads();  // the tile is rewritten by the next unit
  }
}

// out_t[c] (+)= sum_blk partial[blk][t * seg + c] for up to 4 destination segments of equal length `seg`
struct CsArgs { const float* partial; int nblk; int N; int seg; float* out[4]; int accumulate; };
__global__ __launch_bounds__(256) void colsum_finalize_kernel(CsArgs a) {
  __shared__ float red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + tx;
  float s = 0.f;
  if (n < a.N)
    for (int b = ty; b < a.nblk; b += 16) s += a.partial[(long)b * a.N + n];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < a.N) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][tx];
    const int seg = n / a.seg;
    float* o = a.out[seg] + (n - seg * a.seg);
    *o = a.accumulate ? *o + t : t;
  }
}

// Several such column reductions in ONE launch (per layer: the two LayerNorm partial sets -> dgamma, dbeta, bias
// gradient of the producing Linear; the column-sum partials of dqkv -> bq, bk, bv and of du -> b1): these are ~10-us,
// latency-bound launches, four of them per layer on the weight-gradient stream otherwise.  256 threads = 64 columns x 4
// row groups (256-byte contiguous reads per row), fixed summation order (deterministic).
struct CrSet { const float* src; int nrows; int ncols; int seg; float* out[3]; };
struct CrArgs { CrSet set[8]; int n; int accumulate; };
__global__ __launch_bounds__(256) void colreduce_multi_kernel(CrArgs a) {
  __shared__ float red[4][64];
  const CrSet& st = a.set[blockIdx.y];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  if (blockIdx.x * 64 >= st.ncols) return;
  float s0 = 0.f, s1 = 0.f;
  if (n < st.ncols) {
    int b = ty;
    for (; b + 4 < st.nrows; b += 8) {  // two independent chains: more loads in flight
      s0 += st.src[(long)b * st.ncols + n];
      s1 += st.src[(long)(b + 4) * st.ncols + n];
    }
    for (; b < st.nrows; b += 4) s0 += st.src[(long)b * st.ncols + n];
  }
  red[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0 && n < st.ncols) {
    const float t = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    const int seg = n / st.seg;
    float* o = st.out[seg] + (n - seg * st.seg);
    if (st.out[seg]) *o = a.accumulate ? *o + t : t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// grouped NT ping-pong kernel on blocked-transposed operands (structure, hazards and swizzle: gemm.hip gemm3_kernel,
// configuration 256 x 256, one bf16 pass)
// ---------------------------------------------------------------------------------------------------------------
struct DwProb {
  const bf16_raw* a; const bf16_raw* b;  // AT [kt][a_rows][64] at this problem's first row, BT [kt][b_rows][64]
  long ka, kb;                           // elements between consecutive K-tiles (rows_total * 64)
  float* out; const float* mask; long ldo;
  int M, N;                              // output rows (dY columns of this problem) x columns (X columns)
  int tile0, tiles_n;
  // row-major form (TN): a = dY [rows, lda] at this problem's first column, b = X [rows, ldb]; ka / kb = 64 * ld;
  // cs (may be NULL): fp32 [tiles_n, M] partial column sums of dY (the bias gradient; see below)
  long lda, ldb; float* cs;
};
// sk_slots / sk_flags (stream-K form only): caller-owned workspace -- one partial-tile slot (SK_SLOT_FLOATS floats) and one
// hand-off flag (zero between launches) per workgroup
struct DwArgs { DwProb p[MAXT]; int nprob; int nk; int total_tiles; int accumulate; float* sk_slots; unsigned* sk_flags; };
constexpr int SK_SLOT_F4 = 34 * 512;              // float4 per slot: 32 accumulator registers + 2 bias-sum registers per thread
constexpr long SK_SLOT_FLOATS = 4L * SK_SLOT_F4;  // 278 528 bytes

__device__ __forceinline__ int swz3(int row) {
  return ((row >> 1) & 1) | (((row >> 3) & 1) << 1) | ((((row >> 4) ^ (row >> 2)) & 1) << 2);
}

// TN = true: the operands are the ROW-MAJOR activations themselves (dY [rows, N_out], X [rows, K_in], rows = the reduction
// dimension) -- no K-major images, no re-layout pass.  A K-tile is 64 rows; a half-tile is [64 rows][128 columns] (256-B
// rows, 16 KB like the NT half-tile, so staging units, DMA instruction counts and the vmcnt accounting are unchanged);
// MFMA fragments (8 consecutive rows of one column per lane) are gathered with ds_read_b64_tr_b16 -- two 4(row) x
// 16(column) transposing block reads per fragment, the same LDS bytes per MFMA as the NT form.  32-byte column blocks are
// XOR-swizzled with h(row) = (row & 3) | ((row >> 3) & 1) << 2 on the DMA source address and undone by the reads: the 8
// rows a 32-lane half touches hit 8 different 32-byte bank groups.  Needs rows % 64 == 0 and M, N multiples of 8.
// Bias gradients (column sums of dY over the rows) come out of the same pass: the waves with wc == 0 add up the A
// fragments they hold anyway (v_dot2c_f32_bf16 against ones, after the MFMA section) for the K-tiles kt = tn (mod
// tiles_n) of their tile row, so the tiles of a row share the work; partial row tn of cs is summed by vl_colreduce_multi.
// The reads are inline assembly on purpose: the compiler puts s_waitcnt vmcnt(0) in front of the
// __builtin_amdgcn_ds_read_tr16_b64 intrinsic whenever LDS-DMA loads are outstanding (it cannot prove the read does not
// alias them), which drains the whole prefetch pipeline before every fragment section -- measured 515 us instead of
// 330 us for one layer's six problems.  The counted vmcnt waits + barriers of the loop are what orders DMA and reads;
// the results are complete after the s_waitcnt lgkmcnt(0) that opens every MFMA section.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
template <int OFF>
__device__ __forceinline__ bf16x8 tr_frag(unsigned lds_addr) {
  static_assert(OFF >= 0 && OFF + 4 * 256 < 65536, "LDS offset field");
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(lds_addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(lds_addr), "n"(OFF + 4 * 256));
  const u32x4 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
  return __builtin_bit_cast(bf16x8, v);
}
template <bool T, int OFF>
__device__ __forceinline__ bf16x8 dw_frag(unsigned lds_addr) {  // lds_addr: this lane's fragment address in the current stage
  if constexpr (T) return tr_frag<OFF>(lds_addr);
  else return *(const __attribute__((address_space(3))) bf16x8*)(uintptr_t)(lds_addr + OFF);
}
// s + one * the 8 bf16 values (fp32 accumulation); one = bf16 {1, 1} on the K-tiles this tile sums, {0, 0} on the others
// s += one . (dword t of the fragment): two bf16 values, fp32 accumulation (assembly: the dot products have to stay at
// their place inside the MFMA section)
__device__ __forceinline__ void frag_dot(float& s, const bf16x8& f, int t, unsigned one_bits) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 w = __builtin_bit_cast(u32x4, f);
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s) : "s"(one_bits), "v"(w[t]));
}

// MODE bit 0: the A operand (dY) is row-major (transposing reads), else its K-major image; bit 1: the same for X (X).
//
// SK = true, the STREAM-K form (a fixed CU budget for the weight gradients): the grid is G workgroups, G chosen by the
// caller (not by the tile count), and the linearised iteration space {tile} x {K-tile} of the launch is cut into G equal
// contiguous ranges.  A workgroup walks its range as segments (tile, [k0, k0 + len)): a segment that starts a tile
// (k0 == 0) OWNS it -- it runs the epilogue, after adding the partial accumulators of the workgroups that cover the rest of
// the tile's K range, in K order (fixed summation order: deterministic for a given G); a segment with k0 > 0 -- only ever
// the FIRST segment of a workgroup -- stores its raw accumulators (lane-linear, the owner runs the same code) into its
// slot of the caller's workspace and raises its flag.  Owners only wait for workgroups with a HIGHER index, and only for
// their first segment, which waits for nothing: no cycle; G <= the CU count keeps every workgroup resident or next in
// the dispatch order.  Hand-off protocol (per-XCD L2s are not coherent): plain 16-byte stores -> every wave's
// s_waitcnt vmcnt(0) -> workgroup barrier -> lane 0: agent-scope release fence, s_waitcnt vmcnt(0), relaxed agent-scope
// flag store | relaxed agent-scope poll -> agent-scope acquire fence, s_waitcnt vmcnt(0) -> workgroup barrier -> plain
// loads; the owner then clears the flag (flags are zero between launches).
template <int MODE, bool SK>
__global__ __launch_bounds__(512) void dw_grouped_kernel(DwArgs ga) {
  constexpr bool TA = (MODE & 1) != 0, TB = (MODE & 2) != 0;
  constexpr int WC = 4;                        // wave grid 2 (M) x 4 (N)
  constexpr int MI = 4, NJ = 2;                // 16 x 16 MFMA tiles per quadrant
  constexpr int AH = 128, BH = 128;            // rows per half-tile
  constexpr int OFF_A1 = AH * 128, OFF_B0 = 2 * AH * 128, OFF_B1 = OFF_B0 + BH * 128, STAGE = OFF_B1 + BH * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const bool late = wave >= 4;

  // XCD-contiguous virtual workgroup index: neighbours in the work list (tiles that share an operand panel; stream-K:
  // the two sides of a hand-off) sit on one XCD, i.e. behind one L2
  const int nwg = SK ? (int)gridDim.x : ga.total_tiles;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int nk_tile = ga.nk;
  // this workgroup's range of the iteration space
  long it, it_end;
  long sk_per = 0; int sk_rem = 0;
  if (SK) {
    const long total = (long)ga.total_tiles * nk_tile;
    sk_per = total / nwg; sk_rem = (int)(total - sk_per * nwg);
    it = swz * sk_per + (swz < sk_rem ? swz : sk_rem);
    it_end = it + sk_per + (swz < sk_rem ? 1 : 0);
  } else {
    it = (long)swz * nk_tile; it_end = it + nk_tile;
  }
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

 while (it < it_end) {  // segments (SK = false: exactly one, the whole tile)
  // (the lane index is laundered per segment, and again behind the K loop: per-lane address arithmetic is then recomputed
  // where it is used instead of being hoisted out of the segment loop and kept alive across the K loop, which has no
  // register to spare)
  int lane = tid & 63;
  if (SK) asm volatile("" : "+v"(lane));
  const int frow = lane & 15, fk = lane >> 4;
  const int tile_id = SK ? (int)(it / nk_tile) : swz;
  const int k0 = SK ? (int)(it - (long)tile_id * nk_tile) : 0;
  const int nk = SK ? (int)((it_end - it) < (long)(nk_tile - k0) ? (it_end - it) : (long)(nk_tile - k0)) : nk_tile;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAXT; ++i)
    if (i < ga.nprob && tile_id >= ga.p[i].tile0) pi = i;
  const DwProb& P = ga.p[pi];
  const int lt = tile_id - P.tile0;
  const int tm = lt / P.tiles_n, tn = lt - tm * P.tiles_n;
  const int row0 = tm * 256, col0 = tn * 256;

  unsigned soff[4][2];  // per-lane byte offsets from the (uniform) operand base: 32-bit, so the DMA uses the saddr + voffset form
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool isB = x >= 2;
      if (isB ? TB : TA) {
        const int r = 4 * (wave + 8 * j) + (lane >> 4);  // row of the K-tile; unit = 4 rows x 256 B
        const int c = lane & 15;                         // physical 16-byte chunk of the row
        const int lb = (c >> 1) ^ ((r & 3) | (((r >> 3) & 1) << 2));  // logical 32-B block landing at physical c >> 1
        int gcol = (isB ? col0 + (x & 1) * BH : row0 + (x & 1) * AH) + lb * 16 + (c & 1) * 8;
        const int lim = (isB ? P.N : P.M) - 8;
        gcol = gcol < lim ? gcol : lim;  // columns past the edge re-read valid data; their products are never stored
        soff[x][j] = (unsigned)(((long)r * (isB ? P.ldb : P.lda) + gcol) * 2);
      } else {
        const int r = 8 * (wave + 8 * j) + (lane >> 3);
        const int lc = (lane & 7) ^ swz3(r);
        int g = (isB ? col0 + (x & 1) * BH : row0 + (x & 1) * AH) + r;
        const int lim = (isB ? P.N : P.M) - 1;
        g = g < lim ? g : lim;  // rows past the edge re-read a valid row; their products are never stored
        soff[x][j] = (unsigned)((g * 64 + lc * 8) * 2);
      }
    }
  const long kstep[2] = {P.ka * 2, P.kb * 2};  // bytes
  const char* const obase[2] = {(const char*)P.a + (long)k0 * kstep[0], (const char*)P.b + (long)k0 * kstep[1]};
  constexpr int XOFF[4] = {0, OFF_A1, OFF_B0, OFF_B1};
// (SK: the 32-bit per-lane offsets are laundered at every use, so the compiler keeps them as 8 registers instead of 8
// loop-invariant zero-extended 64-bit pairs)
#define DW_ISSUE(x, kt)                                                                                          \
  do {                                                                                                           \
    unsigned o0_ = soff[x][0], o1_ = soff[x][1];                                                                 \
    if (SK) { asm volatile("" : "+v"(o0_)); asm volatile("" : "+v"(o1_)); }                                      \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(obase[(x) >> 1] + (long)(kt) * kstep[(x) >> 1] + o0_),          \
                                     (lds_ptr_t)(smem + ((kt) & 1) * STAGE + XOFF[x] + wave * 1024), 16, 0, 0);  \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(obase[(x) >> 1] + (long)(kt) * kstep[(x) >> 1] + o1_),          \
                                     (lds_ptr_t)(smem + ((kt) & 1) * STAGE + XOFF[x] + (wave + 8) * 1024), 16, 0, 0); \
  } while (0)
#define DW_WAIT(issued)                                                                                          \
  do {                                                                                                           \
    if (issued) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                 \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
  } while (0)

  f32x4 acc[2][2][MI][NJ];
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[a_][b_][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS byte addresses of this lane's fragments in stage 0; they flip to the other stage in place after every K-tile
  unsigned a_o[MI][2], b_o[NJ][2];
  {
    // transposed-read lane geometry: group fk owns rows 8 fk .. 8 fk + 7 of a 32-deep step; inside the group lane
    // 4 q + pp addresses row q, columns 4 pp .. 4 pp + 3 of the 4 x 16 block (the second read is 4 rows below)
    const int q = (lane & 15) >> 2, pp = lane & 3, hq = q | ((fk & 1) << 2);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wr * (MI * 16) + i * 16 + frow;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        a_o[i][kk] = smem_lds + (TA ? (kk * 32 + 8 * fk + q) * 256 + (((wr * MI + i) ^ hq) << 5) + pp * 8
                        : row * 128 + (((4 * kk + fk) ^ swz3(row)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = wc * (NJ * 16) + j * 16 + frow;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        b_o[j][kk] = smem_lds + (TB ? (kk * 32 + 8 * fk + q) * 256 + (((wc * NJ + j) ^ hq) << 5) + pp * 8
                        : row * 128 + (((4 * kk + fk) ^ swz3(row)) << 4));
    }
  }
  float bsum[2][MI];  // column sums of dY for this wave's output rows (bias gradient partials)
#pragma unroll
  for (int qm = 0; qm < 2; ++qm)
#pragma unroll
    for (int i = 0; i < MI; ++i) bsum[qm][i] = 0.f;
  const bool bias_wave = P.cs != nullptr && wc == 0;

  // fa[0]: A fragments of quadrant row 0 -- rows i >= MI / 2 are read one phase EARLY (in the otherwise read-free last
  // phase of the previous K-tile), which levels the LDS reads of the four phases from 12 / 4 / 8 / 0 to 8 / 4 / 8 / 4 fragments
  // (LEVEL = false, the stream-K form: one fragment set for both quadrant rows and no early half -- the segment loop
  // around the K loop leaves no room for the second set; the levelling is worth ~2 % of the kernel)
  constexpr bool LEVEL = !SK;
  constexpr int FQ = LEVEL ? 1 : 0;  // fa[qm * FQ]
  bf16x8 fa[LEVEL ? 2 : 1][MI][2], fb0[NJ][2], fb1[NJ][2];
#define DW_READ_A(st, qm, i0, i1)                                                                                \
  _Pragma("unroll") for (int i = (i0); i < (i1); ++i) {                                                         \
    fa[(qm) * FQ][i][0] = dw_frag<TA, (qm) * OFF_A1>(a_o[i][0]);                                                 \
    fa[(qm) * FQ][i][1] = dw_frag<TA, (qm) * OFF_A1>(a_o[i][1]);                                                 \
  }
#define DW_READ_B(st, qn, fb)                                                                                    \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                              \
    fb[j][0] = dw_frag<TB, OFF_B0 + (qn) * (BH * 128)>(b_o[j][0]);                                   \
    fb[j][1] = dw_frag<TB, OFF_B0 + (qn) * (BH * 128)>(b_o[j][1]);                                   \
  }
  // after the second MFMA section of a quadrant row, on the selected K-tiles: the column sums of its A fragments
  // (v_dot2c_f32_bf16 against {1, 1}: 4 per fragment; consecutive ones are independent)
#define DW_BIAS(qm)                                                                                              \
  do {                                                                                                           \
    if (bias_now) {                                                                                              \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                           \
      _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                              \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) frag_dot(bsum[qm][i], fa[(qm) * FQ][i][kk], t, 0x3F803F80u);           \
    }                                                                                                            \
  } while (0)
#define DW_MFMA(qm, qn, fb)                                                                                \
  do {                                                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                             \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                             \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                             \
        acc[qm][qn][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[(qm) * FQ][i][kk], acc[qm][qn][i][j], 0, 0, 0); \
    }                                                                                                            \
    __builtin_amdgcn_s_setprio(0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
  } while (0)

  // prologue: K-tile 0 complete, A0 / B0 of K-tile 1 in flight
  DW_ISSUE(0, 0); DW_ISSUE(2, 0); DW_ISSUE(3, 0); DW_ISSUE(1, 0);
  if (nk > 1) {
    DW_ISSUE(0, 1); DW_ISSUE(2, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (LEVEL) DW_READ_A(st, 0, MI / 2, MI);  // K-tile 0's early half
  if (late) __builtin_amdgcn_s_barrier();  // stagger

  for (int kt = 0; kt < nk; ++kt) {
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
    const unsigned flip = (kt & 1) ? 0u - (unsigned)STAGE : (unsigned)STAGE;  // to the other stage
    const bool bias_now = bias_wave && ((k0 + kt) % P.tiles_n) == tn;
    DW_READ_B(st, 0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    DW_READ_A(st, 0, 0, LEVEL ? MI / 2 : MI);
    if (n1) DW_ISSUE(3, kt + 1);
    DW_WAIT(n1);
    DW_MFMA(0, 0, fb0);
    DW_READ_B(st, 1, fb1);
#pragma unroll
    for (int j = 0; j < NJ; ++j) { b_o[j][0] += flip; b_o[j][1] += flip; }
    if (n1) DW_ISSUE(1, kt + 1);
    DW_WAIT(n1);
    DW_MFMA(0, 1, fb1);
    DW_BIAS(0);
    DW_READ_A(st, 1, 0, MI);
#pragma unroll
    for (int i = 0; i < MI; ++i) { a_o[i][0] += flip; a_o[i][1] += flip; }
    if (n2) DW_ISSUE(0, kt + 2);
    DW_WAIT(n2);
    DW_MFMA(1, 1, fb1);
    // (A0 of K-tile kt + 1 landed with the wait of the previous phase and is visible after its barriers; the fragment
    // addresses already point to that stage)
    if (LEVEL) DW_READ_A(st, 0, MI / 2, MI);
    if (n2) DW_ISSUE(2, kt + 2);
    DW_WAIT(n2);
    DW_MFMA(1, 0, fb0);
    DW_BIAS(1);
  }
  if (!late) __builtin_amdgcn_s_barrier();
#undef DW_ISSUE
#undef DW_WAIT
#undef DW_READ_A
#undef DW_READ_B
#undef DW_MFMA
#undef DW_BIAS

  if (SK) {
    lane = tid & 63;
    asm volatile("" : "+v"(lane));
  }
  if (SK) {
    if (k0 != 0) {
      // contributor: raw accumulators (+ bias partial sums) -> this workgroup's slot, then the flag.  Register r of
      // thread tid at float4 index r * 512 + tid: every store instruction of a wave is one contiguous KB
      float4* slot = reinterpret_cast<float4*>(ga.sk_slots + (long)swz * SK_SLOT_FLOATS) + (wave * 64 + lane);
#pragma unroll
      for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              const f32x4 v = acc[qm][qn][i][j];
              slot[(((qm * 2 + qn) * MI + i) * NJ + j) * 512] = make_float4(v[0], v[1], v[2], v[3]);
            }
      slot[32 * 512] = make_float4(bsum[0][0], bsum[0][1], bsum[0][2], bsum[0][3]);
      slot[33 * 512] = make_float4(bsum[1][0], bsum[1][1], bsum[1][2], bsum[1][3]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the fence's own wait)
        __hip_atomic_store(ga.sk_flags + swz, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      it += nk;
      continue;
    }
    // owner: add the partial tiles of the workgroups that cover [nk, nk_tile) of this tile, in K order
    int covered = nk;
    for (int c = swz + 1; covered < nk_tile; ++c) {
      if (tid == 0) {
        while (__hip_atomic_load(ga.sk_flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(16);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      const float4* slot = reinterpret_cast<const float4*>(ga.sk_slots + (long)c * SK_SLOT_FLOATS) + (wave * 64 + lane);
      // (two halves of 16 loads each, fenced: all 32 in flight at once would need 128 more registers than there are)
#pragma unroll
      for (int qm = 0; qm < 2; ++qm) {
        float4 v[2][MI][NJ];
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[qn][i][j] = slot[(((qm * 2 + qn) * MI + i) * NJ + j) * 512];
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              acc[qm][qn][i][j][0] += v[qn][i][j].x; acc[qm][qn][i][j][1] += v[qn][i][j].y;
              acc[qm][qn][i][j][2] += v[qn][i][j].z; acc[qm][qn][i][j][3] += v[qn][i][j].w;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const float4 b0 = slot[32 * 512], b1 = slot[33 * 512];
        bsum[0][0] += b0.x; bsum[0][1] += b0.y; bsum[0][2] += b0.z; bsum[0][3] += b0.w;
        bsum[1][0] += b1.x; bsum[1][1] += b1.y; bsum[1][2] += b1.z; bsum[1][3] += b1.w;
      }
      __syncthreads();  // (every thread's loads of the slot have returned: the adds above consumed them)
      if (tid == 0) __hip_atomic_store(ga.sk_flags + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long c_len = sk_per + (c < sk_rem ? 1 : 0);  // workgroup c's range starts at this tile's K-tile `covered`
      covered += (int)(c_len < (long)(nk_tile - covered) ? c_len : (long)(nk_tile - covered));
    }
  }

  if (bias_wave) {  // rows 8 fk .. 8 fk + 7 of every 32-deep step were summed per lane group: add the 4 groups
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        float v = bsum[qm][i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        const int m = row0 + qm * AH + wr * (MI * 16) + i * 16 + (lane & 15);
        if (lane < 16 && m < P.M) P.cs[(long)tn * P.M + m] = v;
      }
  }

  // epilogue: lane & 15 -> output row (dY column), 4 * (lane >> 4) + reg -> 4 consecutive output columns
#pragma unroll
  for (int qm = 0; qm < 2; ++qm)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = row0 + qm * AH + wr * (MI * 16) + i * 16 + (lane & 15);
      if (m >= P.M) continue;
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n0 = col0 + qn * BH + wc * (NJ * 16) + j * 16 + 4 * (lane >> 4);
          if (n0 >= P.N) continue;
          const long o = (long)m * P.ldo + n0;
          f32x4 v = acc[qm][qn][i][j];
          if (n0 + 3 < P.N) {
            if (P.mask) {
              const float4 mk = *reinterpret_cast<const float4*>(P.mask + o);
              v[0] *= mk.x; v[1] *= mk.y; v[2] *= mk.z; v[3] *= mk.w;
            }
            if (ga.accumulate) {
              const float4 old = *reinterpret_cast<const float4*>(P.out + o);
              v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
            }
            *reinterpret_cast<float4*>(P.out + o) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            for (int r = 0; r < 4 && n0 + r < P.N; ++r) {
              float x = v[r];
              if (P.mask) x *= P.mask[o + r];
              if (ga.accumulate) x += P.out[o + r];
              P.out[o + r] = x;
            }
          }
        }
    }
  it += nk;
  if (SK && it < it_end) {  // the epilogue's stores share vmcnt with the next segment's counted DMA waits: drain them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
 }  // segments
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int64_t vl_blocked_elems(int64_t M, int64_t N) { return ((M + 63) / 64) * 64 * N; }

// stream-K workspace: [flags: 256 x 4 B, padded to 4 KB][budget partial-tile slots]
static constexpr int64_t SK_FLAG_BYTES = 4096;
extern "C" int64_t vl_dw_streamk_ws_bytes(int64_t budget) {
  return budget > 0 ? SK_FLAG_BYTES + budget * SK_SLOT_FLOATS * 4 : 0;
}

// tab: HOST array of n x VL_TR_FIELDS int64 {src, ld, N, dst, colsum_partial (0 = none), 0}
extern "C" int vl_transpose_blocked(const int64_t* tab, int64_t n, int64_t M, int64_t max_blocks, void* stream) {
#ifdef VL_EXPERIMENT_SKIP_RELAYOUT  // timing experiment only (wrong results): what the step costs without the re-layout
  if (M > 4096) return 0;
#endif
  VL_CHECK_ARG(tab && n >= 1 && n <= MAXT && M >= 1 && M < (1LL << 31), "vl_transpose_blocked: bad arguments");
  TrArgs a{};
  a.n = (int)n; a.M = (int)M; a.mblocks = (int)((M + 63) / 64);
  int units = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t* t = tab + i * VL_TR_FIELDS;
    TrEntry& e = a.e[i];
    e.src = (const bf16_raw*)t[0]; e.ld = t[1]; e.N = (int)t[2]; e.dst = (bf16_raw*)t[3]; e.colsum = (float*)t[4];
    VL_CHECK_ARG(e.src && e.dst && e.N > 0 && (e.N & 63) == 0 && e.ld >= e.N && (e.ld & 7) == 0 && al16(e.src) && al16(e.dst),
                 "vl_transpose_blocked: entry %d: N must be a multiple of 64, ld a multiple of 8, pointers 16-byte aligned", i);
    e.unit0 = units;
    units += a.mblocks * (e.N >> 6);
  }
  a.total_units = units;
  // fewer workgroups = a gentler stream beside MFMA-bound kernels (each holds 8 KB in flight): the caller picks
  const int cap = max_blocks > 0 ? (int)(max_blocks < 65536 ? max_blocks : 65536) : 4096;
  int grid = units < cap ? units : cap;
  hipLaunchKernelGGL(transpose_blocked_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  VL_CHECK_LAUNCH("vl_transpose_blocked");
  return 0;
}

extern "C" int vl_colsum_finalize(const float* partial, int64_t nblk, int64_t N, float* const* outs, int64_t nout,
                                  int accumulate, void* stream) {
  VL_CHECK_ARG(partial && outs && nblk >= 1 && N >= 1 && nout >= 1 && nout <= 4 && N % nout == 0,
               "vl_colsum_finalize: bad arguments");
  CsArgs a{};
  a.partial = partial; a.nblk = (int)nblk; a.N = (int)N; a.seg = (int)(N / nout); a.accumulate = accumulate;
  for (int i = 0; i < nout; ++i) {
    VL_CHECK_ARG(outs[i], "vl_colsum_finalize: null destination %d", i);
    a.out[i] = outs[i];
  }
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, (hipStream_t)stream, a);
  VL_CHECK_LAUNCH("vl_colsum_finalize");
  return 0;
}

// tab: HOST array of n (<= 4) x VL_CR_FIELDS int64 {src, nrows, ncols, seg, out0, out1, out2, 0}: out_t[c] (+)= sum_rows
// src[row][t * seg + c] (a NULL out_t skips that segment)
extern "C" int vl_colreduce_multi(const int64_t* tab, int64_t n, int accumulate, void* stream) {
  VL_CHECK_ARG(tab && n >= 1 && n <= 8, "vl_colreduce_multi: bad arguments");
  CrArgs a{};
  a.n = (int)n; a.accumulate = accumulate;
  int maxc = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t* t = tab + i * VL_CR_FIELDS;
    CrSet& st = a.set[i];
    st.src = (const float*)t[0]; st.nrows = (int)t[1]; st.ncols = (int)t[2]; st.seg = (int)t[3];
    VL_CHECK_ARG(st.src && st.nrows >= 1 && st.ncols >= 1 && st.seg >= 1 && st.ncols % st.seg == 0 && st.ncols / st.seg <= 3,
                 "vl_colreduce_multi: set %d: bad sizes", i);
    for (int k = 0; k < 3; ++k) st.out[k] = (float*)t[4 + k];
    if (st.ncols > maxc) maxc = st.ncols;
  }
  hipLaunchKernelGGL(colreduce_multi_kernel, dim3((unsigned)((maxc + 63) / 64), (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
  VL_CHECK_LAUNCH("vl_colreduce_multi");
  return 0;
}

// probs: HOST array of nprob x VL_DW_FIELDS int64 {aT, a_rows_total, bT, b_rows_total, out, ldo, mask (0 = none), M, N, 0}
// (rowmajor: {dY, lda, X, ldb, out, ldo, mask, M, N, colsum partials [ceil(N / 256), M] or 0})
template <int MODE, bool SK>
static int dw_launch_t(const DwArgs& a, int grid, const char* fn, void* stream) {
  const size_t lds = 2 * 65536;
  static bool attr_set = false;  // per instantiation; idempotent, so a race only repeats the call
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_grouped_kernel<MODE, SK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vl_set_error(-3, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((dw_grouped_kernel<MODE, SK>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
  VL_CHECK_LAUNCH(fn);
  return 0;
}

// budget > 0: the stream-K form on `budget` workgroups with the caller's workspace `ws` (vl_dw_streamk_ws_bytes(budget)
// bytes, zero-filled once by the caller; launches that share it must be ordered on one stream)
static int dw_launch(const int64_t* probs, int64_t nprob, int64_t K, int accumulate, int mode, int64_t budget, void* ws,
                     int64_t ws_bytes, void* stream) {
  const bool rowmajor = mode != 0;
  const char* fn = budget > 0 ? "vl_dw_grouped_streamk" : rowmajor ? "vl_dw_grouped_rowmajor" : "vl_dw_grouped";
  VL_CHECK_ARG(mode >= 0 && mode <= 3, "%s: mode must be 0..3", fn);
  VL_CHECK_ARG(probs && nprob >= 1 && nprob <= MAXT && K >= 1, "%s: bad arguments", fn);
  VL_CHECK_ARG(!rowmajor || K % 64 == 0, "%s: the row count must be a multiple of 64 (got %lld)", fn, (long long)K);
  DwArgs a{};
  a.nprob = (int)nprob; a.nk = (int)((K + 63) / 64); a.accumulate = accumulate;
  int tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const int64_t* t = probs + i * VL_DW_FIELDS;
    DwProb& p = a.p[i];
    // per operand: row-major -> (pointer, leading dimension), K-major image -> (pointer, columns of the image); the
    // distance between K-tiles is 64 * that number either way
    p.a = (const bf16_raw*)t[0]; p.ka = t[1] * 64; p.b = (const bf16_raw*)t[2]; p.kb = t[3] * 64;
    p.lda = t[1]; p.ldb = t[3];
    p.out = (float*)t[4]; p.ldo = t[5]; p.mask = (const float*)t[6]; p.M = (int)t[7]; p.N = (int)t[8];
    p.cs = (float*)t[9];
    VL_CHECK_ARG(p.a && p.b && p.out && p.M > 0 && p.N > 0 && t[1] >= p.M && t[3] >= p.N && p.ldo >= p.N,
                 "%s: problem %d: bad pointers / sizes", fn, i);
    VL_CHECK_ARG(al16(p.a) && al16(p.b) && al16(p.out) && al16(p.mask) && (p.ldo & 3) == 0,
                 "%s: problem %d: pointers must be 16-byte aligned, ldo a multiple of 4", fn, i);
    VL_CHECK_ARG(!(mode & 1) || ((p.M & 7) == 0 && p.M >= 8 && (t[1] & 7) == 0),
                 "%s: problem %d: M and lda must be multiples of 8 for a row-major dY", fn, i);
    VL_CHECK_ARG(!(mode & 2) || ((p.N & 7) == 0 && p.N >= 8 && (t[3] & 7) == 0),
                 "%s: problem %d: N and ldb must be multiples of 8 for a row-major X", fn, i);
    p.tile0 = tiles;
    p.tiles_n = (p.N + 255) / 256;
    tiles += ((p.M + 255) / 256) * p.tiles_n;
  }
  a.total_tiles = tiles;
  if (budget > 0) {
    VL_CHECK_ARG(budget <= 256, "%s: the workgroup budget must be <= 256 (one 8-wave workgroup per CU; owners wait for resident workgroups)", fn);
    VL_CHECK_ARG(ws && al16(ws) && ws_bytes >= vl_dw_streamk_ws_bytes(budget), "%s: workspace missing / too small / unaligned", fn);
    // a range of less than two K-tiles is not worth a hand-off: shrink the grid (the partition stays valid for any grid)
    int64_t g = budget;
    const int64_t total = (int64_t)tiles * a.nk;
    if (g > total / 2) g = total / 2 > 0 ? total / 2 : 1;
    a.sk_flags = reinterpret_cast<unsigned*>(ws);
    a.sk_slots = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + SK_FLAG_BYTES);
    switch (mode) {
      case 0: return dw_launch_t<0, true>(a, (int)g, fn, stream);
      case 1: return dw_launch_t<1, true>(a, (int)g, fn, stream);
      case 2: return dw_launch_t<2, true>(a, (int)g, fn, stream);
      default: return dw_launch_t<3, true>(a, (int)g, fn, stream);
    }
  }
  switch (mode) {
    case 0: return dw_launch_t<0, false>(a, tiles, fn, stream);
    case 1: return dw_launch_t<1, false>(a, tiles, fn, stream);
    case 2: return dw_launch_t<2, false>(a, tiles, fn, stream);
    default: return dw_launch_t<3, false>(a, tiles, fn, stream);
  }
}

extern "C" int vl_dw_grouped(const int64_t* probs, int64_t nprob, int64_t K, int accumulate, void* stream) {
  return dw_launch(probs, nprob, K, accumulate, 0, 0, nullptr, 0, stream);
}
extern "C" int vl_dw_grouped_rowmajor(const int64_t* probs, int64_t nprob, int64_t rows, int accumulate, void* stream) {
  return dw_launch(probs, nprob, rows, accumulate, 3, 0, nullptr, 0, stream);
}
// mode bit 0: dY row-major (else its K-major image: fields 0 / 1 = image, image columns), bit 1: the same for X
extern "C" int vl_dw_grouped_mixed(const int64_t* probs, int64_t nprob, int64_t rows, int accumulate, int mode, void* stream) {
  return dw_launch(probs, nprob, rows, accumulate, mode, 0, nullptr, 0, stream);
}
// The stream-K form (see dw_grouped_kernel): the same products on a FIXED number of workgroups -- `budget` CUs of the chip
// for the weight gradients, the rest stays with whatever runs beside them -- each walking an equal share of the launch's
// (tile, K-tile) iterations; tiles cut by a share boundary are completed by their owner from the partial tiles in `ws`.
// Results are deterministic for a given budget (fixed summation order), and differ from the one-workgroup-per-tile form
// by fp32 re-association only.
extern "C" int vl_dw_grouped_streamk(const int64_t* probs, int64_t nprob, int64_t rows, int accumulate, int mode,
                                     int64_t budget, void* ws, int64_t ws_bytes, void* stream) {
  VL_CHECK_ARG(budget >= 1, "vl_dw_grouped_streamk: budget must be >= 1");
  return dw_launch(probs, nprob, rows, accumulate, mode, budget, ws, ws_bytes, stream);
}
