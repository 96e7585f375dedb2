// int4 dequant-GEMM for gfx950, throughput version: out[M, N] = x[M, K] (fp16) . W[K, N].
//
// Weight format ("k-packed, pair order"): qw[K/8][N] uint32, one word = 8 consecutive k of one
// column, nibbles ordered k0,k2,k4,k6,k1,k3,k5,k7 -- exactly what the reference's gptq_shuffle
// writes (csrc/quantization/gptq/qdq_4.cuh:12-29).  In that order four mask-and-or operations
// turn one word into the 8 fp16 values (k0..k7) of a column, i.e. into one 16-byte MFMA B operand
// held by the lane that loaded the word: the weights never pass through LDS.  AWQ checkpoints are
// brought into the same format once by vla_awq_repack (AWQ packs along N, dequantize.cuh:41-61).
// zeros: qz[K/G][N/8] uint32, natural nibble order; effective zero = nibble + ZOFF (GPTQ stores
// zero - 1, q_gemm.cu:1413; AWQ stores the zero itself).  scales: [K/G][N] fp16.
//
// Each weight is reconstructed bit-exactly as the reference does it -- (q - z) exactly in fp16,
// then one fp16 multiply by the scale (gemm_kernels.cu:156-179, q_gemm.cu:939-947) -- and the
// contraction runs on 16x16x32 fp16 MFMAs with fp32 accumulation, rounded to fp16 once.
//
// Workgroup: 512 threads = 8 waves as 4 (M) x 2 (N), tile 256 x 128, K step 64.  x is staged
// through LDS in full 128-byte rows (XOR-swizzled 16-byte chunks, conflict-free ds_read_b128),
// double buffered, one barrier per K step; weight words, zeros and scales go global -> registers.
// Small-N layers are split along K into fp32 slabs that a second kernel sums.
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <type_traits>

#include "vla_common.h"

// Register-ring depths of the small-batch and grouped launches (kernel studies,
// tools/exp/build_ring.sh): deeper rings were measured and lose everywhere -- weight words 7
// steps ahead instead of 3: layer GEMMs at batch 16 58.5 -> 64.0 us, at batch 64 85.3 -> 121.8
// (the 1024-thread launches spill), Mixtral block unchanged within 2 %; x 8 steps ahead on top:
// 69.1 / 131.2.  Loads retire in order, so a wave can use x(t + 4) only after every older weight
// request has landed: the distance that counts is the shortest one in the stream.
#ifndef VLA_Q4_WD
#define VLA_Q4_WD 4    // weight ring depth of the small-batch and grouped launches (4 or 8)
#endif
#ifndef VLA_Q4_XA
#define VLA_Q4_XA 4    // x sets (steps of x in flight) of 32- and 64-row tiles (4 or 8)
#endif
#ifndef VLA_Q4_XAG
#define VLA_Q4_XAG 2   // x sets of grouped 128-row tiles (2 or 4)
#endif

namespace vla {
namespace {

constexpr int TM = 256, TN = 128, TK = 64;
constexpr int kThreadsG = 512;
constexpr int X_TILE_BYTES = TM * TK * 2;  // 32 KiB per buffer

__device__ __forceinline__ f16x2 as_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ uint32_t as_u32(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }

template <int ZOFF, bool SPLIT>
__global__ __launch_bounds__(kThreadsG, 2) void q4_gemm_fast_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * X_TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * TN, m0 = blockIdx.y * TM;
  const int k_begin = blockIdx.z * k_per_split;
  const int k_end = min(k, k_begin + k_per_split);

  // ---- x staging: thread -> (row = 64 i + tid / 8, 16-byte chunk tid % 8)
  const int srow = tid >> 3, schunk = tid & 7;
  const _Float16* xsrc = x + (int64_t)(m0 + srow) * k + 8 * schunk;
  const int sdst = srow * 128 + ((schunk ^ (srow & 7)) << 4);
  u32x4 xr[4];
  auto load_x = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xr[i] = (m0 + srow + 64 * i < m)
                  ? *reinterpret_cast<const u32x4*>(xsrc + (int64_t)(64 * i) * k + k0)
                  : u32x4{0, 0, 0, 0};
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<u32x4*>(xs + buf * X_TILE_BYTES + sdst + i * (64 * 128)) = xr[i];
  };

  // ---- weight words: lane (g, c) owns column ncol[b] and rows k8 = .. + g of each 32-deep step
  int ncol[4];
  bool nok[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    ncol[b] = n0 + 64 * wn + 16 * b + c;
    nok[b] = ncol[b] < n;
  }
  uint32_t wr[2][4];
  auto load_w = [&](int k0, uint32_t (&w)[2][4]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int64_t row = (int64_t)((k0 + 32 * kk) / 8 + g) * n;
#pragma unroll
      for (int b = 0; b < 4; ++b) w[kk][b] = nok[b] ? qw[row + ncol[b]] : 0u;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-group dequantisation constants of this lane's 4 columns
  f16x2 zc1[4], zc2[4], sc2[4];
  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};

  load_x(k_begin);
  load_w(k_begin, wr);
  store_x(0);
  __syncthreads();

  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += TK) {
    const bool has_next = k0 + TK < k_end;
    uint32_t wnx[2][4];
    if (has_next) {
      load_x(k0 + TK);
      load_w(k0 + TK, wnx);
    }
    const unsigned char* xb = xs + buf * X_TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ks = k0 + 32 * kk;
      if (ks == k_begin || ks % group_size == 0) {  // wave-uniform: new quantisation group
        const int grp = ks / group_size;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          uint32_t zw = 0;
          _Float16 s = (_Float16)0.f;
          if (nok[b]) {
            zw = qz[(int64_t)grp * (n >> 3) + (ncol[b] >> 3)];
            s = scales[(int64_t)grp * n + ncol[b]];
          }
          const int z = (int)((zw >> (4 * (ncol[b] & 7))) & 15u) + ZOFF;
          const _Float16 z1 = (_Float16)(float)(-(1024 + z)), z2 = (_Float16)(float)(-(64 + z));
          zc1[b] = f16x2{z1, z1};
          zc2[b] = f16x2{z2, z2};
          sc2[b] = f16x2{s, s};
        }
      }
      // B operands: dequantise this lane's words
      u32x4 bf[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t w = wr[kk][b], w8 = w >> 8;
        const f16x2 d0 = as_h2((w & 0x000f000fu) | 0x64006400u) + zc1[b];
        const f16x2 d1 = __builtin_elementwise_fma(as_h2((w & 0x00f000f0u) | 0x64006400u),
                                                   sixteenth, zc2[b]);
        const f16x2 d2 = as_h2((w8 & 0x000f000fu) | 0x64006400u) + zc1[b];
        const f16x2 d3 = __builtin_elementwise_fma(as_h2((w8 & 0x00f000f0u) | 0x64006400u),
                                                   sixteenth, zc2[b]);
        bf[b][0] = as_u32(d0 * sc2[b]);
        bf[b][1] = as_u32(d1 * sc2[b]);
        bf[b][2] = as_u32(d2 * sc2[b]);
        bf[b][3] = as_u32(d3 * sc2[b]);
      }
      // A operands from LDS, 16 MFMAs
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int row = 64 * wm + 16 * a + c;
        const u32x4 af = *reinterpret_cast<const u32x4*>(
            xb + row * 128 + (((4 * kk + g) ^ (row & 7)) << 4));
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af),
                                                             __builtin_bit_cast(f16x8, bf[b]),
                                                             acc[a][b], 0, 0, 0);
      }
    }
    if (has_next) {
      store_x(buf ^ 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int b = 0; b < 4; ++b) wr[kk][b] = wnx[kk][b];
    }
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue: lane (g, c) holds column ncol[b], rows 4g .. 4g+3 of each 16-row tile
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 64 * wm + 16 * a + 4 * g + i;
      if (row >= m) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (!nok[b]) continue;
        if constexpr (SPLIT)
          reinterpret_cast<float*>(outp)[((int64_t)blockIdx.z * m + row) * n + ncol[b]] = acc[a][b][i];
        else
          reinterpret_cast<_Float16*>(outp)[(int64_t)row * n + ncol[b]] = (_Float16)acc[a][b][i];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Main kernel: 32x32x16 MFMAs, wave tile 128 (M) x 32 (N), workgroup = NW waves side by side
// along N, i.e. a 128 x 32*NW output tile, K step 64.  Every wave owns its columns outright, so
// no weight is dequantised twice in a workgroup, and per K step a wave spends 52 vector ops on
// its 4 words per lane for 16 MFMAs of 32 cycles.  A 128-row tile also halves the bytes of x a
// workgroup pulls from L2 per flop against a 256-row tile of the same area (x is re-read by every
// column tile; the int4 weights are the small operand).
// Lane l: c = l & 31 is its column (B) / row within a 32-row tile (A), h = l >> 5 picks k 8h..8h+7
// of a 16-deep slice.  Weight words (with their group's zeros and scales) are fetched three K
// steps ahead into a four-name register ring, x two steps ahead into two register sets and parked
// in the LDS double buffer one step ahead; A fragments are read one slice ahead of their MFMAs.
// Requires group_size % 64 == 0.
// LDS x image: row r (128 bytes = 8 chunks of 16), chunk q stored at slot q ^ ((r >> 1) & 7):
// with rows on lanes this keeps every ds_read_b128 lane group on 16 distinct 16-byte slots.
// MT = 32-row tiles per workgroup: 4 for the batch the kernel is built around; 1 and 2 for small
// batches, where the work is reading the weights once (HBM-bound) and a 128-row tile would spend
// MFMA passes and x traffic on rows that do not exist.

// Store a wave's transposed accumulator tiles: acc[a][4 j + e] = out[row0 + 32 a][col0 + 8 j + e]
// (row0 / col0 already include the lane's c and 4 h); fp32 into slab `row_shift / m` of a split
// product, fp16 otherwise.  n % 8 == 0, so a quad of columns is inside the matrix or outside it.
template <bool SPLIT, int MT>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[MT], void* __restrict__ outp, int row0,
                                           int m, int col0, int n, int64_t row_shift) {
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int row = row0 + 32 * a;
    if (row >= m) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + 8 * j;
      if (col >= n) continue;
      const int64_t at = (row_shift + row) * n + col;
      if constexpr (SPLIT) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outp) + at) =
            f32x4{acc[a][4 * j], acc[a][4 * j + 1], acc[a][4 * j + 2], acc[a][4 * j + 3]};
      } else {
        const f16x2 lo = {(_Float16)acc[a][4 * j], (_Float16)acc[a][4 * j + 1]};
        const f16x2 hi = {(_Float16)acc[a][4 * j + 2], (_Float16)acc[a][4 * j + 3]};
        *reinterpret_cast<u32x2*>(reinterpret_cast<_Float16*>(outp) + at) = u32x2{as_u32(lo), as_u32(hi)};
      }
    }
  }
}

// GROUPED (several weight sets in one launch, e.g. the experts of an MoE block): x and out hold
// the row blocks of all sets back to back, gp.row_offsets[e] .. [e + 1] on the device says which
// rows belong to set e, blockIdx.y = e * tiles_per_set + row tile, and a workgroup whose tile
// lies past its set's rows leaves at once.  (Skipping the empty 32-row sub-tiles of a partly
// filled tile behind wave-uniform branches was measured and dropped: the branches around the
// fragment reads and MFMAs cost the schedule more than the skipped work saves, 227 + 115 ->
// 257 + 136 us for the two launches of a Mixtral-8x7B block at 256 tokens.)
struct Q4Groups {
  const int* row_offsets;      // [num_sets + 1]
  const int* x_rows;           // null, or [total_rows]: row r of the product reads x[x_rows[r]]
  int tiles_per_set;
  int total_rows;              // rows of x / out, the slab pitch of a split product
  int64_t w_stride, z_stride, s_stride;   // elements between consecutive sets
};

// KW (small batches): the K range of a workgroup is cut into KW parts, each worked on by its own
// set of NW waves with its own x tiles in LDS; the parts' accumulators are summed through LDS at
// the end (a fixed tree, so the result does not depend on timing).  A 32-row step is ~300 cycles
// of issue work behind a few hundred cycles of load and MFMA-chain latency, so what a small-batch
// launch lacks is waves per SIMD -- and more K slices across workgroups would pay for them with
// fp32 slabs in HBM.
// LV: how many of the tile's MT 32-row sub-tiles hold rows.  It is MT everywhere except in a
// grouped launch, where a set's share of a decode batch rarely fills its tile: the kernel looks
// the row count up and runs the body built for that many live sub-tiles, which spends no x load,
// no fragment read and no MFMA on the empty ones.  (The same skip as wave-uniform branches inside
// one body was measured first: it pays at 16-64 tokens per block but costs full tiles their
// schedule, Mixtral-8x7B block at 256 tokens 0.425 -> 0.46 ms.)
template <int ZOFF, bool SPLIT, int NW, int MT, bool GROUPED, bool G2, int KW, int LV>
__device__ __forceinline__ void q4_gemm_m32_body(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split, const Q4Groups& gp,
    unsigned char* __restrict__ xs_all) {
  constexpr int THREADS = 64 * NW;               // threads of one K part
  constexpr int TMW = 32 * MT;                   // rows per workgroup
  constexpr int X2_TILE_BYTES = TMW * TK * 2;    // one LDS buffer: 16 KiB at MT = 4
  static_assert((TMW * 8) % THREADS == 0, "x staging covers the tile in whole passes");
  static_assert(KW == 1 || (NW <= 4 && (KW == 2 || KW == 4)), "LDS reduction fits the x buffers");
  static_assert(LV >= 1 && LV <= MT, "live sub-tiles");
  constexpr int XROWS = THREADS / 8;        // rows covered by one pass of the staging threads
  // 16-byte chunks of the x tile per thread: the staging passes that reach a live sub-tile
  constexpr int XCH = (32 * LV + XROWS - 1) / XROWS;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave % NW, wk = wave / NW;
  const int tid = threadIdx.x - wk * THREADS;    // thread index within the K part
  unsigned char* xs = xs_all + wk * (2 * X2_TILE_BYTES);
  const int h = lane >> 5, c = lane & 31;
  int tile_y = blockIdx.y, row_base = 0, slab_rows = m;
  if constexpr (GROUPED) {
    const int set = blockIdx.y / gp.tiles_per_set;
    tile_y = blockIdx.y % gp.tiles_per_set;
    row_base = gp.row_offsets[set];
    m = gp.row_offsets[set + 1] - row_base;
    slab_rows = gp.total_rows;
    if (tile_y * TMW >= m) return;
    if (!gp.x_rows) x += (int64_t)row_base * k;
    qw += (int64_t)set * gp.w_stride;
    qz += (int64_t)set * gp.z_stride;
    scales += (int64_t)set * gp.s_stride;
  }
  const int n0 = blockIdx.x * (32 * NW), m0 = tile_y * TMW;
  const int k_part = k_per_split / KW;           // the host keeps this a whole number of groups
  const int k_begin = blockIdx.z * k_per_split + wk * k_part;
  const int k_end = min(k, k_begin + k_part);
  const int nsteps = k_part / TK;                // the same for every part: barriers stay in step
  if (nsteps <= 0) return;
#ifdef VLA_Q4_STAMP   // diagnostic build: where a workgroup's time goes (100 MHz ticks)
  const uint64_t st_entry = __builtin_amdgcn_s_memrealtime();
  uint64_t st_first = 0;
#endif
  auto phys = [&](int t) { return min(t, nsteps - 1); };   // past the end: re-read the last step

  // ---- x staging: thread -> (row = XROWS i + tid / 8, chunk tid % 8).  Rows past m are clamped
  // to the last row: they only feed accumulator rows that are never stored.
  const int srow = tid >> 3, schunk = tid & 7;
  const _Float16* xsrc[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    int r = min(m0 + srow + XROWS * i, m - 1);
    if constexpr (GROUPED) {
      if (gp.x_rows) r = gp.x_rows[row_base + r];
    }
    xsrc[i] = x + (int64_t)r * k + 8 * schunk;
  }
  struct XSet { u32x4 v[XCH]; };
  auto load_x = [&](int step, XSet& d) {
    const int k0 = k_begin + phys(step) * TK;
#pragma unroll
    for (int i = 0; i < XCH; ++i) d.v[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + k0);
  };
  auto store_x = [&](int buf, const XSet& d) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int r = srow + XROWS * i;
      *reinterpret_cast<u32x4*>(xs + buf * X2_TILE_BYTES + r * 128 +
                                ((schunk ^ ((r >> 1) & 7)) << 4)) = d.v[i];
    }
  };

  // ---- weights: lane (h, c) owns column ncol; per 16-deep slice s one word, row k0/8 + 2s + h
  const int ncol = n0 + 32 * wn + c;
  const bool nok = ncol < n;
  const int ncl = nok ? ncol : 0;                    // clamped: loads stay in bounds, result unused
  const uint32_t* wsrc = qw + (int64_t)h * n + ncl;
  const uint32_t* zsrc = qz + (ncl >> 3);
  // scales are fetched as the aligned 32-bit word holding columns (ncl & ~1, ncl | 1): a 16-bit
  // load would be widened by an ALU instruction that has to wait for it
  const uint32_t* ssrc = reinterpret_cast<const uint32_t*>(scales) + (ncl >> 1);
  const int zshift = 4 * (ncl & 7);
  const uint32_t ssel = (ncl & 1) ? 0x03020302u : 0x01000100u;
  // The zero word and the scale pair word of a K step travel with its weight words, except when
  // a group is exactly two K steps (G2, group size 128): then they are fetched once per group
  // into a two-name ring of their own, two steps before the group's first step -- the two small
  // loads per step cost as much issue time as the weight words themselves.
  struct ZS { uint32_t z; uint32_t s; };
  struct WStep { uint32_t w[4]; ZS zs; };
  const int last_group = (k_end - 1) / group_size;
  auto load_zs = [&](int grp, ZS& d) {
    grp = min(grp, last_group);
    d.z = zsrc[(int64_t)grp * (n >> 3)];
    d.s = ssrc[(int64_t)grp * (n >> 1)];
  };
  auto load_w = [&](int step, WStep& d) {
    const int k0 = k_begin + phys(step) * TK;
    const uint32_t* p = wsrc + (int64_t)(k0 >> 3) * n;
    // An unsliced small-batch launch (K parts inside the workgroup only: the widest layers) reads
    // every weight word exactly once: nontemporal there (gate_up 4096 x 28672 at batch 1: 22.5 ->
    // 19.3 us).  Sliced launches lose with the hint (qkv at batch 128: 22.7 -> 26.2 us).
    constexpr bool WNT = !SPLIT && KW == 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr (WNT)
        d.w[s] = __builtin_nontemporal_load(p + (int64_t)(2 * s) * n);
      else
        d.w[s] = p[(int64_t)(2 * s) * n];
    }
    if constexpr (!G2) load_zs(k0 / group_size, d.zs);
  };

  // a single 32-row tile would chain all four MFMAs of a step on one accumulator (each waits for
  // the one before): odd slices get a second one there
  constexpr bool DUAL = LV == 1;
  f32x16 acc[LV], acc_odd[1];
#pragma unroll
  for (int a = 0; a < LV; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  if constexpr (DUAL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_odd[0][i] = 0.f;
  }

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  // (w & mask) | 0x6400.. as ONE v_and_or_b32: the instruction takes a single scalar or literal
  // operand, so the exponent pattern has to sit in a vector register the compiler cannot fold
  // back into a literal (it would split the operation into v_and + v_or)
  uint32_t c1024 = 0x64006400u;
  asm volatile("" : "+v"(c1024));
  constexpr uint32_t kLoMask = 0x000f000fu, kHiMask = 0x00f000f0u;
  auto and_or = [](uint32_t a, uint32_t m, uint32_t c) { return (a & m) | c; };

  // One K step t: issue the loads of x(t+XAHEAD) into `xnew` and of the weights of step t+3 into
  // `wnew`, compute step t from `wcur` and LDS buffer t & 1, then park x(t+1) (`xpark`) in the
  // other buffer.  Register sets rotate by NAME (the loop is unrolled by four), so a register
  // that a load is still writing is never copied.
  // Loads complete in program order.  With two x sets (128-row tiles) x(t+1) was requested a
  // step ago, behind the weights of step t+1: parking it waits for those too, a step before
  // they are needed -- harmless where a step is ~1000 cycles long.  Small row tiles have short
  // steps, so they keep four x sets: x(t+1) was requested three steps ago, ahead of every
  // weight load that step t does not need anyway, and the weights get their full three steps.
  constexpr int XAHEAD = MT <= 2 ? VLA_Q4_XA : (GROUPED ? VLA_Q4_XAG : 2);
  // Weight ring: the words of a step are requested WD - 1 steps ahead.
  constexpr int WD = (MT <= 2 || GROUPED) ? VLA_Q4_WD : 4;
  static_assert(WD == 4 || WD == 8, "ring depth");
  // G2: zcur holds the constants of step t's group; on even steps znew receives those of the
  // group WD - 2 steps ahead (the name the previous group has just released).
  auto k_step = [&](int t, const WStep& wcur, WStep& wnew, XSet& xnew, const XSet& xpark,
                    const ZS& zcur, ZS& znew) {
    const int buf = t & 1;
    load_x(t + XAHEAD, xnew);
    load_w(t + WD - 1, wnew);
    if constexpr (G2) {
      if ((t & 1) == 0)
        load_zs((k_begin + phys(min(t + WD - 2, nsteps - 2)) * TK) / group_size, znew);
    }
    __builtin_amdgcn_sched_barrier(0);               // keep the loads at the top of the step

    // -(1024 + z) and -(64 + z) as fp16 bit patterns (ulp is 1 resp. 1/16 there, z <= 16)
    const ZS& zs = G2 ? zcur : wcur.zs;
    const uint32_t z = ((zs.z >> zshift) & 15u) + ZOFF;
    const f16x2 zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);
    const f16x2 zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);
    const f16x2 sc2 = as_h2(__builtin_amdgcn_perm(zs.s, zs.s, ssel));

    const unsigned char* xb = xs + buf * X2_TILE_BYTES;
    auto read_a = [&](int s, u32x4 (&af)[LV]) {
#pragma unroll
      for (int a = 0; a < LV; ++a) {
        const int row = 32 * a + c;
        af[a] = *reinterpret_cast<const u32x4*>(
            xb + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4));
      }
    };
    auto dequant = [&](int s) {
      const uint32_t w = wcur.w[s], w8 = w >> 8;
      const f16x2 d0 = as_h2(and_or(w, kLoMask, c1024)) + zc1;
      const f16x2 d1 = __builtin_elementwise_fma(as_h2(and_or(w, kHiMask, c1024)), sixteenth, zc2);
      const f16x2 d2 = as_h2(and_or(w8, kLoMask, c1024)) + zc1;
      const f16x2 d3 = __builtin_elementwise_fma(as_h2(and_or(w8, kHiMask, c1024)), sixteenth, zc2);
      u32x4 bf;
      bf[0] = as_u32(d0 * sc2);
      bf[1] = as_u32(d1 * sc2);
      bf[2] = as_u32(d2 * sc2);
      bf[3] = as_u32(d3 * sc2);
      return bf;
    };
    auto mma = [&](const u32x4 (&af)[LV], const u32x4& bf) {
#pragma unroll
      for (int a = 0; a < LV; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf),
                                                        __builtin_bit_cast(f16x8, af[a]), acc[a], 0,
                                                        0, 0);
    };
    auto mma_odd = [&](const u32x4 (&af)[LV], const u32x4& bf) {
      if constexpr (DUAL)
        acc_odd[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf),
                                                            __builtin_bit_cast(f16x8, af[0]),
                                                            acc_odd[0], 0, 0, 0);
      else
        mma(af, bf);
    };
    // fragments of slice s+1 are requested before the MFMAs of slice s are issued
#define VLA_SB __builtin_amdgcn_sched_barrier(0)
    u32x4 fa[LV], fb[LV];
    read_a(0, fa);
    u32x4 b0 = dequant(0);
    VLA_SB;
    read_a(1, fb);
    VLA_SB;
    mma(fa, b0);
    u32x4 b1 = dequant(1);
    VLA_SB;
    read_a(2, fa);
    VLA_SB;
    mma_odd(fb, b1);
    b0 = dequant(2);
    VLA_SB;
    read_a(3, fb);
    VLA_SB;
    mma(fa, b0);
    b1 = dequant(3);
    VLA_SB;
    mma_odd(fb, b1);
#undef VLA_SB
    if (t + 1 < nsteps) store_x(buf ^ 1, xpark);
    __syncthreads();
#ifdef VLA_Q4_STAMP
    if (t == 0) st_first = __builtin_amdgcn_s_memrealtime();
#endif
  };

  // Register rings, rotated by NAME: the loop is unrolled by the weight ring's depth and every
  // index below is a constant after unrolling.  x(j) lives in set j % XAHEAD, the weights of step
  // j in set j % WD.
  WStep wr[WD];
  XSet xr[XAHEAD];
  constexpr int ZD = WD / 2;       // G2: the constants of group group0 + i live in zr[i % ZD]
  ZS zr[ZD];
#pragma unroll
  for (int j = 0; j < XAHEAD; ++j) load_x(j, xr[j]);
  if constexpr (G2) {
#pragma unroll
    for (int j = 0; j < ZD - 1; ++j)
      load_zs((k_begin + phys(min(2 * j, nsteps - 1)) * TK) / group_size, zr[j]);
  }
#pragma unroll
  for (int j = 0; j < WD - 1; ++j) load_w(j, wr[j]);
  store_x(0, xr[0]);
  __syncthreads();
  auto step_u = [&](auto uc, int t) {
    constexpr int u = decltype(uc)::value;
    constexpr int zi = (u >> 1) % ZD;
    k_step(t, wr[u % WD], wr[(u + WD - 1) % WD], xr[u % XAHEAD], xr[(u + 1) % XAHEAD], zr[zi],
           zr[(zi + ZD - 1) % ZD]);
  };
  using std::integral_constant;
  int t = 0;
  for (; t + WD <= nsteps; t += WD) {
    step_u(integral_constant<int, 0>{}, t);
    step_u(integral_constant<int, 1>{}, t + 1);
    step_u(integral_constant<int, 2>{}, t + 2);
    step_u(integral_constant<int, 3>{}, t + 3);
    if constexpr (WD == 8) {
      step_u(integral_constant<int, 4>{}, t + 4);
      step_u(integral_constant<int, 5>{}, t + 5);
      step_u(integral_constant<int, 6>{}, t + 6);
      step_u(integral_constant<int, 7>{}, t + 7);
    }
  }
  if (t < nsteps) step_u(integral_constant<int, 0>{}, t);
  if (t + 1 < nsteps) step_u(integral_constant<int, 1>{}, t + 1);
  if (t + 2 < nsteps) step_u(integral_constant<int, 2>{}, t + 2);
  if constexpr (WD == 8) {
    if (t + 3 < nsteps) step_u(integral_constant<int, 3>{}, t + 3);
    if (t + 4 < nsteps) step_u(integral_constant<int, 4>{}, t + 4);
    if (t + 5 < nsteps) step_u(integral_constant<int, 5>{}, t + 5);
    if (t + 6 < nsteps) step_u(integral_constant<int, 6>{}, t + 6);
  }

#ifdef VLA_Q4_STAMP
  const uint64_t st_loop1 = __builtin_amdgcn_s_memrealtime();
#endif
  if constexpr (DUAL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] += acc_odd[0][i];
  }
  // ---- the K parts of the workgroup: parts [stride, 2 stride) hand their sums to parts
  // [0, stride) through LDS (the x buffers, free after the last step's barrier), stride = KW/2 .. 1
  if constexpr (KW > 1) {
    float* red = reinterpret_cast<float*>(xs_all);
    constexpr int SLOT = NW * LV * 16 * 64;        // floats one part hands over
#pragma unroll
    for (int stride = KW / 2; stride >= 1; stride >>= 1) {
      if (wk >= stride && wk < 2 * stride) {
        float* dst = red + (wk - stride) * SLOT + wn * (LV * 16 * 64) + lane;
#pragma unroll
        for (int a = 0; a < LV; ++a)
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[(a * 16 + i) * 64] = acc[a][i];
      }
      __syncthreads();
      if (wk < stride) {
        const float* src = red + wk * SLOT + wn * (LV * 16 * 64) + lane;
#pragma unroll
        for (int a = 0; a < LV; ++a)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[a][i] += src[(a * 16 + i) * 64];
      }
      if (stride > 1) __syncthreads();             // the slots are rewritten by the next round
    }
    if (wk != 0) return;
  }
  // ---- epilogue.  The weights are the MFMA's A operand and x its B operand, so the tile comes
  // out transposed: lane (h, c) holds row 32 a + c of x and, in registers 4 j .. 4 j + 3, the four
  // adjacent columns 8 j + 4 h .. + 3 of the wave's 32 -- one 8- or 16-byte store per quad instead
  // of four 2- or 4-byte ones (the store tail of a 128-row tile was a sixth of the kernel).
  store_tile<SPLIT, LV>(acc, outp, m0 + c, m, n0 + 32 * wn + 4 * h, n,
                        SPLIT ? (int64_t)blockIdx.z * slab_rows + row_base : (int64_t)row_base);
#ifdef VLA_Q4_STAMP
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t st_end = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && ((blockIdx.x + 5 * blockIdx.y + 11 * blockIdx.z) % 61) == 0)
      printf("stamp32 wg %d,%d,%d steps %d entry %llu first +%llu loop1 +%llu end +%llu\n", blockIdx.x,
             blockIdx.y, blockIdx.z, nsteps, (unsigned long long)(st_entry % 100000000ull),
             (unsigned long long)(st_first - st_entry), (unsigned long long)(st_loop1 - st_entry),
             (unsigned long long)(st_end - st_entry));
  }
#endif
}

template <int ZOFF, bool SPLIT, int NW, int MT, bool GROUPED = false, bool G2 = false, int KW = 1>
__global__ __launch_bounds__(64 * NW * KW) void q4_gemm_m32_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split, Q4Groups gp = Q4Groups{}) {
  __shared__ __attribute__((aligned(16))) unsigned char xs_all[KW * 2 * (32 * MT) * TK * 2];
#define VLA_Q4_BODY(LV_)                                                                          \
  q4_gemm_m32_body<ZOFF, SPLIT, NW, MT, GROUPED, G2, KW, LV_>(x, qw, qz, scales, outp, m, n, k,    \
                                                              group_size, k_per_split, gp, xs_all)
  if constexpr (!GROUPED) {
    VLA_Q4_BODY(MT);
  } else {
    const int set = blockIdx.y / gp.tiles_per_set, tile_y = blockIdx.y % gp.tiles_per_set;
    const int rows = gp.row_offsets[set + 1] - gp.row_offsets[set] - tile_y * (32 * MT);
    if (rows <= 0) return;
    const int live = __builtin_amdgcn_readfirstlane((rows + 31) >> 5);
    if constexpr (MT >= 4) {
      if (live >= 4) { VLA_Q4_BODY(MT >= 4 ? 4 : MT); return; }
      if (live == 3) { VLA_Q4_BODY(MT >= 4 ? 3 : MT); return; }
    }
    if constexpr (MT >= 2) {
      if (live >= 2) { VLA_Q4_BODY(MT >= 2 ? 2 : MT); return; }
    }
    VLA_Q4_BODY(1);
  }
#undef VLA_Q4_BODY
}

// ---------------------------------------------------------------------------------------------
// 128-row tiles with every operand brought in by LDS-DMA (global_load_lds): x, the weight words
// and the per-group zeros / scales of a K step land in one LDS stage of a ring, nothing is staged
// through registers, and the only vector-memory wait is one counted s_waitcnt per K step.
// The tile shape, the dequantisation and the MFMA order are those of q4_gemm_m32_kernel at MT = 4.
//
// An LDS-DMA instruction writes 64 lanes x 16 (or 4) bytes contiguously at a wave-uniform LDS
// address (M0); what each lane fetches is free.  So the images are lane-linear and any swizzle is
// applied to the SOURCE address:
//   x    128 rows x 128 bytes, chunk q of row r in slot q ^ ((r >> 1) & 7) (as the register-staged
//        kernel); one instruction = 8 rows, wave w fills rows [16 w .. ) (NW = 8) or [32 w .. ) (NW = 4).
//        Every wave reads all of it: it is ordered by the barrier that closes a step.
//   w    per wave, its own 32 columns: 8 k8-rows x 32 words = one instruction
//   zs   per wave: the 16 scale-pair words and the 4 zero words of its 32 columns (one 4-byte
//        instruction; its other lanes repeat the first 16)
// A wave's weights, zeros and scales are its own: they are ordered by its own vmcnt wait alone, so
// it reads the words of tile t + 1 into registers BEFORE the barrier that closes step t and starts
// step t + 1 with its first dequantisation instead of an LDS round trip.
// Every wave issues PER_TILE instructions per tile, which is what makes one vmcnt literal right.
// Ring: tile t + D is requested during step t (behind the first MFMAs) into the stage step t - 1
// has finished reading (every wave passed that step's barrier after its last fragment read
// returned); at the end of step t a wave waits until at most D - 1 tiles of its own requests are
// outstanding, then the barrier makes the x of tile t + 1 visible to every wave.
//
// One asm statement = the instructions of one tile: M0 is the compiler's register, so it is saved
// and restored around them once.  x and w: 64-bit scalar base + 32-bit per-lane byte offset (the
// lane part never changes, the bases walk along K in scalar registers); zs: per-lane 64-bit
// address (scales and zeros are different tensors), stepped once per quantisation group.
template <int XP>
__device__ __forceinline__ void glds_tile(const uint32_t (&xoff)[XP], uint64_t xbase, uint32_t xdst,
                                          uint32_t woff, uint64_t wbase, uint32_t wdst,
                                          const void* zsrc, uint32_t zdst) {
  static_assert(XP == 2 || XP == 4, "x instructions per wave");
  unsigned keep;
  if constexpr (XP == 2) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %8\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %8\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %9\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dword %6, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(xoff[0]), "v"(xoff[1]), "s"(xdst), "v"(woff), "s"(wdst), "v"(zsrc), "s"(zdst),
          "s"(xbase), "s"(wbase)
        : "memory");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %10\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %10\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %10\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %10\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %11\n\t"
        "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dword %8, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(xoff[0]), "v"(xoff[1]), "v"(xoff[2]), "v"(xoff[3]), "s"(xdst), "v"(woff), "s"(wdst),
          "v"(zsrc), "s"(zdst), "s"(xbase), "s"(wbase)
        : "memory");
  }
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifndef VLA_Q4_ABL
#define VLA_Q4_ABL 0   // diagnostic builds (wrong results, timing only): 1 no per-step wait and barrier, 2 no dequantisation, 4 no DMA in the loop, 16 no per-step wait, 32 no per-step barrier
#endif
template <int NW, int STAGES>
struct Q4Dma {
  static constexpr int XB = 128 * 128;          // x tile
  static constexpr int WB = 1024 * NW;          // 1 KiB of weight words per wave
  static constexpr int ZB = 256 * NW;           // 256 bytes of scales / zeros per wave
  static constexpr int STAGE = XB + WB + ZB;
  static constexpr int BYTES = STAGES * STAGE;
  static constexpr int XP = 16 / NW;            // x instructions per wave and tile
  static constexpr int PER_TILE = XP + 2;
};

template <int ZOFF, bool SPLIT, int NW, int STAGES, int D>
__global__ __launch_bounds__(64 * NW) void q4_gemm_dma_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split) {
  using L = Q4Dma<NW, STAGES>;
  constexpr int MT = 4;
  static_assert(NW == 4 || NW == 8, "whole x instructions per wave");
  static_assert(D >= 1 && STAGES >= D + 1 && STAGES <= 5, "ring depth");
  static_assert(L::PER_TILE * (D - 1) < 64, "vmcnt is six bits");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // the only LDS object: offset 0

#ifdef VLA_Q4_STAMP
  const uint64_t st_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int n0 = blockIdx.x * (32 * NW), m0 = blockIdx.y * 128;
  const int k_begin = blockIdx.z * k_per_split;
  const int nsteps = min(k_per_split, k - k_begin) / TK;
  if (nsteps <= 0) return;
  const int wcol0 = n0 + 32 * wave;                        // first column of this wave

  // ---- what this lane fetches (columns past n: clamped to the wave's first or to column 0; the
  // products they feed are never stored)
  uint32_t xoff[L::XP];
#pragma unroll
  for (int j = 0; j < L::XP; ++j) {
    const int r = (wave * L::XP + j) * 8 + (lane >> 3);
    const int q = (lane & 7) ^ ((r >> 1) & 7);
    xoff[j] = (uint32_t)(min(m0 + r, m - 1) * k + 8 * q) * 2u;       // M K < 2^30 (host check)
  }
  uint32_t wlane;
  {
    const int r = lane >> 3, col = wcol0 + 4 * (lane & 7);             // k8-row of the step, 4 columns
    wlane = (uint32_t)(r * n + (col < n ? col : 0)) * 4u;
  }
  const unsigned char* zsrc;                               // walks one group per `zs_stride`
  uint32_t zs_stride;
  if ((lane & 31) < 16 || (lane & 31) >= 20) {             // scale pairs (lanes 20.. repeat them)
    const int col = wcol0 + 2 * (lane & 15);
    zsrc = reinterpret_cast<const unsigned char*>(scales) + (size_t)(col < n ? col : 0) * 2;
    zs_stride = (uint32_t)n * 2u;
  } else {                                                 // lanes 16..19 of each half: zero words
    const int col = wcol0 + 8 * (lane & 3);
    zsrc = reinterpret_cast<const unsigned char*>(qz) + (size_t)((col < n ? col : 0) >> 3) * 4;
    zs_stride = (uint32_t)(n >> 3) * 4u;
  }
  zsrc += (size_t)(k_begin / group_size) * zs_stride;
  // scalar bases of the next tile to request; tiles are requested in order 0, 1, 2, ...
  uint64_t xbase = reinterpret_cast<uint64_t>(x) + (uint64_t)k_begin * 2u;
  uint64_t wbase = reinterpret_cast<uint64_t>(qw) + (uint64_t)(k_begin >> 3) * (uint64_t)n * 4u;
  const uint64_t wstep = (uint64_t)(TK / 8) * (uint64_t)n * 4u;
  const int steps_per_group = group_size / TK;
  int group_left = steps_per_group - (k_begin / TK) % steps_per_group;   // tiles left in the group
  const uint32_t xdst0 = (uint32_t)(wave * L::XP * 1024), wdst0 = (uint32_t)(L::XB + wave * 1024),
                 zdst0 = (uint32_t)(L::XB + L::WB + wave * 256);
  auto issue = [&](int stage) {
    const uint32_t base = (uint32_t)(stage * L::STAGE);
    glds_tile<L::XP>(xoff, xbase, base + xdst0, wlane, wbase, base + wdst0, zsrc, base + zdst0);
    xbase += TK * 2;
    wbase += wstep;
    if (--group_left == 0) {
      group_left = steps_per_group;
      zsrc += zs_stride;
    }
  };

  // ---- what this lane reads: column c of the wave, k 8h .. 8h + 7 of each 16-deep slice
  const int woff = L::XB + wave * 1024 + h * 128 + c * 4;  // + 256 s
  const int soff = L::XB + L::WB + wave * 256 + 4 * (c >> 1), zoff = L::XB + L::WB + wave * 256 + 64 + 4 * (c >> 3);
  const int zshift = 4 * (c & 7);
  const uint32_t ssel = (c & 1) ? 0x03020302u : 0x01000100u;
  const int aoff = c * 128;                                // + a 4096 + ((2 s + h) ^ cs) << 4
  const int cs = (c >> 1) & 7;

  f32x16 acc[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  uint32_t c1024 = 0x64006400u;
  asm volatile("" : "+v"(c1024));
  constexpr uint32_t kLoMask = 0x000f000fu, kHiMask = 0x00f000f0u;
  auto and_or = [](uint32_t a, uint32_t msk, uint32_t cc) { return (a & msk) | cc; };

  struct Words { uint32_t w[4], sw, zw; };                 // a tile's raw words of this lane
  struct Consts { f16x2 zc1, zc2, sc2; };
  auto load_words = [&](const unsigned char* st, Words& d) {
#pragma unroll
    for (int s = 0; s < 4; ++s) d.w[s] = *reinterpret_cast<const uint32_t*>(st + woff + 256 * s);
    d.sw = *reinterpret_cast<const uint32_t*>(st + soff);
    d.zw = *reinterpret_cast<const uint32_t*>(st + zoff);
  };
  auto make_consts = [&](const Words& d) {
    const uint32_t z = ((d.zw >> zshift) & 15u) + ZOFF;
    Consts r;
    r.zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);
    r.zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);
    r.sc2 = as_h2(__builtin_amdgcn_perm(d.sw, d.sw, ssel));
    return r;
  };
  auto read_a = [&](const unsigned char* st, int s, u32x4 (&af)[MT]) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
      af[a] = *reinterpret_cast<const u32x4*>(st + aoff + a * 4096 + (((2 * s + h) ^ cs) << 4));
  };
  auto dequant = [&](uint32_t wv, const Consts& q) {
    const uint32_t w8 = wv >> 8;
    if constexpr ((VLA_Q4_ABL & 2) != 0) return u32x4{wv & 0x3bff3bffu, w8 & 0x3bff3bffu, wv & 0x37ff37ffu, w8 & 0x33ff33ffu};
    const f16x2 d0 = as_h2(and_or(wv, kLoMask, c1024)) + q.zc1;
    const f16x2 d1 = __builtin_elementwise_fma(as_h2(and_or(wv, kHiMask, c1024)), sixteenth, q.zc2);
    const f16x2 d2 = as_h2(and_or(w8, kLoMask, c1024)) + q.zc1;
    const f16x2 d3 = __builtin_elementwise_fma(as_h2(and_or(w8, kHiMask, c1024)), sixteenth, q.zc2);
    u32x4 bf;
    bf[0] = as_u32(d0 * q.sc2);
    bf[1] = as_u32(d1 * q.sc2);
    bf[2] = as_u32(d2 * q.sc2);
    bf[3] = as_u32(d3 * q.sc2);
    return bf;
  };
  auto mma = [&](const u32x4 (&af)[MT], const u32x4& bf) {
#pragma unroll
    for (int rep = 0; rep < (((VLA_Q4_ABL & 64) != 0) ? 2 : 1); ++rep)
#pragma unroll
    for (int a = 0; a < MT; ++a)
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf),
                                                      __builtin_bit_cast(f16x8, af[a]), acc[a], 0, 0,
                                                      0);
  };
#define VLA_SB __builtin_amdgcn_sched_barrier(0)
  Words wd;                                                // words of the tile about to be computed
  auto step = [&](auto stage_c, int t) {
    constexpr int ST = decltype(stage_c)::value;
    const unsigned char* st = lds + ST * L::STAGE;
    const unsigned char* stn = lds + ((ST + 1) % STAGES) * L::STAGE;
    const Consts q = make_consts(wd);
    u32x4 fa[MT], fb[MT];
    read_a(st, 0, fa);
    u32x4 b0 = dequant(wd.w[0], q);
    VLA_SB;
    read_a(st, 1, fb);
    VLA_SB;
    mma(fa, b0);
    u32x4 b1 = dequant(wd.w[1], q);
    VLA_SB;
    if constexpr ((VLA_Q4_ABL & 4) == 0) {                 // behind the first MFMAs, not before them
      if (t + D < nsteps) issue((ST + D) % STAGES);
    }
    VLA_SB;
    read_a(st, 2, fa);
    VLA_SB;
    mma(fb, b1);
    b0 = dequant(wd.w[2], q);
    VLA_SB;
    read_a(st, 3, fb);
    VLA_SB;
    mma(fa, b0);
    b1 = dequant(wd.w[3], q);
    VLA_SB;
    mma(fb, b1);
    VLA_SB;
    if constexpr ((VLA_Q4_ABL & 1) == 0) {
      if constexpr ((VLA_Q4_ABL & 16) == 0) {
        if (t + D < nsteps) wait_vmcnt<L::PER_TILE * (D - 1)>(); else wait_vmcnt<0>();
      } else {
        if (t + 1 >= nsteps) wait_vmcnt<0>();
      }
      if (t + 1 < nsteps) load_words(stn, wd);             // this wave's own bytes: no barrier needed
      if constexpr ((VLA_Q4_ABL & 32) == 0) __builtin_amdgcn_s_barrier();
    } else {
      if (t + 1 >= nsteps) wait_vmcnt<0>();
      if (t + 1 < nsteps) load_words(stn, wd);
    }
    asm volatile("" ::: "memory");
  };
#undef VLA_SB

  // ---- prologue: tiles 0 .. D - 1 requested, tile 0 landed
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < nsteps) issue(j);
  if (D <= nsteps) wait_vmcnt<L::PER_TILE * (D - 1)>(); else wait_vmcnt<0>();
  load_words(lds, wd);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#ifdef VLA_Q4_STAMP   // diagnostic build: shader clock of the K loop (cycles per 100 MHz tick)
  const uint64_t st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1 % STAGES>;
  using I2 = std::integral_constant<int, 2 % STAGES>;
  using I3 = std::integral_constant<int, 3 % STAGES>;
  using I4 = std::integral_constant<int, 4 % STAGES>;
  {
    int t = 0;
    for (; t + STAGES <= nsteps; t += STAGES) {
      step(I0{}, t);
      step(I1{}, t + 1);
      if constexpr (STAGES >= 3) step(I2{}, t + 2);
      if constexpr (STAGES >= 4) step(I3{}, t + 3);
      if constexpr (STAGES >= 5) step(I4{}, t + 4);
    }
    if (t < nsteps) step(I0{}, t);
    if (t + 1 < nsteps) step(I1{}, t + 1);
    if constexpr (STAGES >= 4) {
      if (t + 2 < nsteps) step(I2{}, t + 2);
    }
    if constexpr (STAGES >= 5) {
      if (t + 3 < nsteps) step(I3{}, t + 3);
    }
  }
#ifdef VLA_Q4_STAMP
  const uint64_t st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
#endif

  store_tile<SPLIT, MT>(acc, outp, m0 + c, m, wcol0 + 4 * h, n,
                        SPLIT ? (int64_t)blockIdx.z * m : (int64_t)0);
#ifdef VLA_Q4_STAMP
  {
    wait_vmcnt<0>();
    const uint64_t st_end = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x == 0 || threadIdx.x == 64 * NW - 64) && blockIdx.z == 0 && ((blockIdx.x + 3 * blockIdx.y) % 37) == 0)
      printf("stamp wg %d,%d wave %d steps %d entry %llu loop0 +%llu loop1 +%llu end +%llu MHz %.0f cyc/step %.0f\n",
             blockIdx.x, blockIdx.y, wave, nsteps, (unsigned long long)(st_entry % 100000000ull),
             (unsigned long long)(st_r0 - st_entry), (unsigned long long)(st_r1 - st_entry),
             (unsigned long long)(st_end - st_entry),
             100.0 * (double)(st_c1 - st_c0) / (double)(st_r1 - st_r0), (double)(st_c1 - st_c0) / nsteps);
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// The LDS-DMA kernel with 128 x 64 wave tiles: 8 waves = 4 column waves x 2 K parts.  A wave owns
// 64 columns (two 32-column groups, 128 accumulator registers) over HALF of the workgroup's K
// range; the two parts' sums meet through LDS at the end.  Per K step a wave now issues 32 MFMAs
// behind the same 16 fragment reads, one barrier and one round of requests that served 16 before:
// a step of the 128 x 32 tile spends ~600 cycles per SIMD on everything that is not an MFMA (counter
// passes: 143 other instructions per 16 MFMAs, one issue slot each), and measured with the MFMAs
// of a step simply issued twice the matrix pipe's share rose from 0.62 to 0.79.
// The workgroup tile (128 x 256), the grid and the K slicing across workgroups stay as they are.
// Each K part has its own x ring (its 4 waves fill and read it); the barrier is the workgroup's.
// The weight words are requested nontemporal: an unsliced launch reads each of them once per row
// tile and nothing else will (gate_up at batch 256: 72.7 -> 69.5 us).
__device__ __forceinline__ void glds_tile2(const uint32_t (&xoff)[4], uint64_t xbase, uint32_t xdst,
                                           const uint32_t (&woff)[2], uint64_t wbase, uint32_t wdst,
                                           const void* zsrc, uint32_t zdst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %11\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %11\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %11\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %11\n\t"
      "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %12 nt\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %12 nt\n\t"
      "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dword %9, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(xoff[0]), "v"(xoff[1]), "v"(xoff[2]), "v"(xoff[3]), "s"(xdst), "v"(woff[0]), "v"(woff[1]),
        "s"(wdst), "v"(zsrc), "s"(zdst), "s"(xbase), "s"(wbase)
      : "memory");
}

// The same requests in three groups (x pieces 0-1, x pieces 2-3, the weight and zero / scale
// pieces), for issuing them between different MFMA groups of a step (-DVLA_Q4_SPREAD: a study of
// where in the step the requests cost least).
__device__ __forceinline__ void glds_x2(uint32_t xo0, uint32_t xo1, uint64_t xbase, uint32_t xdst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(xo0), "v"(xo1), "s"(xdst), "s"(xbase)
      : "memory");
}
__device__ __forceinline__ void glds_wz2(const uint32_t (&woff)[2], uint64_t wbase, uint32_t wdst,
                                         const void* zsrc, uint32_t zdst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %6 nt\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6 nt\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %4, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(woff[0]), "v"(woff[1]), "s"(wdst), "v"(zsrc), "s"(zdst), "s"(wbase)
      : "memory");
}

template <int STAGES>
struct Q4Dma2 {
  static constexpr int XB = 128 * 128;          // x tile of one K part
  static constexpr int WB = 4 * 2048;           // 2 KiB of weight words per wave
  static constexpr int ZB = 4 * 256;            // 256 bytes of scales / zeros per wave
  static constexpr int PART = XB + WB + ZB;     // one K part's share of a stage
  static constexpr int STAGE = 2 * PART;
  static constexpr int BYTES = STAGES * STAGE;
  static constexpr int PER_TILE = 7;            // 4 x + 2 w + 1 zeros / scales
  static_assert(BYTES >= 4 * 128 * 64 * 4, "the K parts' sums meet in the ring's memory");
};

template <int ZOFF, bool SPLIT, int STAGES, int D>
__global__ __launch_bounds__(512) void q4_gemm_dma2_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split) {
  using L = Q4Dma2<STAGES>;
  constexpr int MT = 4, NT = 2;
  static_assert(D >= 1 && STAGES >= D + 1 && STAGES <= 3 && L::PER_TILE * (D - 1) < 64, "ring depth");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // the only LDS object: offset 0

#ifdef VLA_Q4_STAMP
  const uint64_t st_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave & 3, wk = wave >> 2;                 // column wave, K part
  const int h = lane >> 5, c = lane & 31;
  const int n0 = blockIdx.x * 256, m0 = blockIdx.y * 128;
  const int k_part = k_per_split / 2;                      // the host keeps this whole groups
  const int k_begin = blockIdx.z * k_per_split + wk * k_part;
  const int nsteps = k_part / TK;                          // the same for both parts
  if (nsteps <= 0) return;
  const int wcol0 = n0 + 64 * wn;                          // first column of this wave

  // ---- what this lane fetches (columns past n: clamped to column 0, never stored)
  uint32_t xoff[4], wlane[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wn * 4 + j) * 8 + (lane >> 3);
    const int q = (lane & 7) ^ ((r >> 1) & 7);
    xoff[j] = (uint32_t)(min(m0 + r, m - 1) * k + 8 * q) * 2u;       // M K < 2^30 (host check)
  }
#pragma unroll
  for (int g = 0; g < NT; ++g) {
    const int r = lane >> 3, col = wcol0 + 32 * g + 4 * (lane & 7);
    wlane[g] = (uint32_t)(r * n + (col < n ? col : 0)) * 4u;
  }
  const unsigned char* zsrc;                               // walks one group per `zs_stride`
  uint32_t zs_stride;
  if (lane < 32 || lane >= 40) {                           // scale pairs (lanes 40.. repeat them)
    const int col = wcol0 + 2 * (lane & 31);
    zsrc = reinterpret_cast<const unsigned char*>(scales) + (size_t)(col < n ? col : 0) * 2;
    zs_stride = (uint32_t)n * 2u;
  } else {                                                 // lanes 32..39: zero words
    const int col = wcol0 + 8 * (lane & 7);
    zsrc = reinterpret_cast<const unsigned char*>(qz) + (size_t)((col < n ? col : 0) >> 3) * 4;
    zs_stride = (uint32_t)(n >> 3) * 4u;
  }
  zsrc += (size_t)(k_begin / group_size) * zs_stride;
  uint64_t xbase = reinterpret_cast<uint64_t>(x) + (uint64_t)k_begin * 2u;
  uint64_t wbase = reinterpret_cast<uint64_t>(qw) + (uint64_t)(k_begin >> 3) * (uint64_t)n * 4u;
  const uint64_t wstep = (uint64_t)(TK / 8) * (uint64_t)n * 4u;
  const int steps_per_group = group_size / TK;
  int group_left = steps_per_group - (k_begin / TK) % steps_per_group;
  const uint32_t part0 = (uint32_t)(wk * L::PART);
  const uint32_t xdst0 = part0 + (uint32_t)(wn * 4096), wdst0 = part0 + (uint32_t)(L::XB + wn * 2048),
                 zdst0 = part0 + (uint32_t)(L::XB + L::WB + wn * 256);
  auto issue = [&](int stage) {
    const uint32_t base = (uint32_t)(stage * L::STAGE);
    glds_tile2(xoff, xbase, base + xdst0, wlane, wbase, base + wdst0, zsrc, base + zdst0);
    xbase += TK * 2;
    wbase += wstep;
    if (--group_left == 0) {
      group_left = steps_per_group;
      zsrc += zs_stride;
    }
  };
  // the same tile's requests in three groups (see glds_x2): part 0 and 1 = x, part 2 = the rest
  [[maybe_unused]] auto issue_part = [&](int stage, int part) {
    const uint32_t base = (uint32_t)(stage * L::STAGE);
    if (part == 0) {
      glds_x2(xoff[0], xoff[1], xbase, base + xdst0);
    } else if (part == 1) {
      glds_x2(xoff[2], xoff[3], xbase, base + xdst0 + 0x800);
      xbase += TK * 2;
    } else {
      glds_wz2(wlane, wbase, base + wdst0, zsrc, base + zdst0);
      wbase += wstep;
      if (--group_left == 0) {
        group_left = steps_per_group;
        zsrc += zs_stride;
      }
    }
  };

  // ---- what this lane reads
  const int pbase = wk * L::PART;
  const int woff = pbase + L::XB + wn * 2048 + h * 128 + c * 4;     // + 1024 g + 256 s
  const int zsb = pbase + L::XB + L::WB + wn * 256;
  const int soff = zsb + 4 * (c >> 1), zoff = zsb + 128 + 4 * (c >> 3);   // + 64 g, + 16 g
  const int zshift = 4 * (c & 7);
  const uint32_t ssel = (c & 1) ? 0x03020302u : 0x01000100u;
  const int aoff = pbase + c * 128;                        // + a 4096 + ((2 s + h) ^ cs) << 4
  const int cs = (c >> 1) & 7;

  f32x16 acc[NT][MT];
#pragma unroll
  for (int g = 0; g < NT; ++g)
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[g][a][i] = 0.f;

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  uint32_t c1024 = 0x64006400u;
  asm volatile("" : "+v"(c1024));
  constexpr uint32_t kLoMask = 0x000f000fu, kHiMask = 0x00f000f0u;
  auto and_or = [](uint32_t a, uint32_t msk, uint32_t cc) { return (a & msk) | cc; };

  struct Words { uint32_t w[NT][4], sw[NT], zw[NT]; };     // a tile's raw words of this lane
  struct Consts { f16x2 zc1, zc2, sc2; };
  auto load_words = [&](const unsigned char* st, Words& d) {
#pragma unroll
    for (int g = 0; g < NT; ++g) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        d.w[g][s] = *reinterpret_cast<const uint32_t*>(st + woff + 1024 * g + 256 * s);
      d.sw[g] = *reinterpret_cast<const uint32_t*>(st + soff + 64 * g);
      d.zw[g] = *reinterpret_cast<const uint32_t*>(st + zoff + 16 * g);
    }
  };
  auto make_consts = [&](const Words& d, int g) {
    const uint32_t z = ((d.zw[g] >> zshift) & 15u) + ZOFF;
    Consts r;
    r.zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);
    r.zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);
    r.sc2 = as_h2(__builtin_amdgcn_perm(d.sw[g], d.sw[g], ssel));
    return r;
  };
  auto read_a = [&](const unsigned char* st, int s, u32x4 (&af)[MT]) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
      af[a] = *reinterpret_cast<const u32x4*>(st + aoff + a * 4096 + (((2 * s + h) ^ cs) << 4));
  };
  auto dequant = [&](uint32_t wv, const Consts& q) {
    const uint32_t w8 = wv >> 8;
    const f16x2 d0 = as_h2(and_or(wv, kLoMask, c1024)) + q.zc1;
    const f16x2 d1 = __builtin_elementwise_fma(as_h2(and_or(wv, kHiMask, c1024)), sixteenth, q.zc2);
    const f16x2 d2 = as_h2(and_or(w8, kLoMask, c1024)) + q.zc1;
    const f16x2 d3 = __builtin_elementwise_fma(as_h2(and_or(w8, kHiMask, c1024)), sixteenth, q.zc2);
    u32x4 bf;
    bf[0] = as_u32(d0 * q.sc2);
    bf[1] = as_u32(d1 * q.sc2);
    bf[2] = as_u32(d2 * q.sc2);
    bf[3] = as_u32(d3 * q.sc2);
    return bf;
  };
  auto mma2 = [&](const u32x4 (&af)[MT], const u32x4& bf0, const u32x4& bf1) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      acc[0][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf0),
                                                         __builtin_bit_cast(f16x8, af[a]), acc[0][a],
                                                         0, 0, 0);
      acc[1][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf1),
                                                         __builtin_bit_cast(f16x8, af[a]), acc[1][a],
                                                         0, 0, 0);
    }
  };
#define VLA_SB __builtin_amdgcn_sched_barrier(0)
  Words wd;                                                // words of the tile about to be computed
  auto step = [&](auto stage_c, int t) {
    constexpr int ST = decltype(stage_c)::value;
    const unsigned char* st = lds + ST * L::STAGE;
    const unsigned char* stn = lds + ((ST + 1) % STAGES) * L::STAGE;
    const Consts q0 = make_consts(wd, 0), q1 = make_consts(wd, 1);
    u32x4 fa[MT], fb[MT];
    read_a(st, 0, fa);
    u32x4 p0 = dequant(wd.w[0][0], q0), p1 = dequant(wd.w[1][0], q1);
    VLA_SB;
    read_a(st, 1, fb);
    VLA_SB;
    mma2(fa, p0, p1);
    u32x4 r0 = dequant(wd.w[0][1], q0), r1 = dequant(wd.w[1][1], q1);
    VLA_SB;
#ifdef VLA_Q4_SPREAD
    if (t + D < nsteps) issue_part((ST + D) % STAGES, 2);  // weights first: they come from HBM
#else
    if (t + D < nsteps) issue((ST + D) % STAGES);          // behind the first MFMAs, not before them
#endif
    VLA_SB;
    read_a(st, 2, fa);
    VLA_SB;
    mma2(fb, r0, r1);
    p0 = dequant(wd.w[0][2], q0);
    p1 = dequant(wd.w[1][2], q1);
    VLA_SB;
#ifdef VLA_Q4_SPREAD
    if (t + D < nsteps) issue_part((ST + D) % STAGES, 0);
    VLA_SB;
#endif
    read_a(st, 3, fb);
    VLA_SB;
    mma2(fa, p0, p1);
    r0 = dequant(wd.w[0][3], q0);
    r1 = dequant(wd.w[1][3], q1);
    VLA_SB;
#ifdef VLA_Q4_SPREAD
    if (t + D < nsteps) issue_part((ST + D) % STAGES, 1);
    VLA_SB;
#endif
    mma2(fb, r0, r1);
    VLA_SB;
    if (t + D < nsteps) wait_vmcnt<L::PER_TILE * (D - 1)>(); else wait_vmcnt<0>();
    if (t + 1 < nsteps) load_words(stn, wd);               // this wave's own bytes: no barrier needed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#undef VLA_SB

#ifdef VLA_Q4_PRIO   // study: static priority for the second-dispatched half of the workgroup
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  // ---- prologue: tiles 0 .. D - 1 requested, tile 0 landed
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < nsteps) issue(j);
  if (D <= nsteps) wait_vmcnt<L::PER_TILE * (D - 1)>(); else wait_vmcnt<0>();
  load_words(lds, wd);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#ifdef VLA_Q4_STAMP
  const uint64_t st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1 % STAGES>;
  using I2 = std::integral_constant<int, 2 % STAGES>;
  {
    int t = 0;
    for (; t + STAGES <= nsteps; t += STAGES) {
      step(I0{}, t);
      step(I1{}, t + 1);
      if constexpr (STAGES >= 3) step(I2{}, t + 2);
    }
    if (t < nsteps) step(I0{}, t);
    if (t + 1 < nsteps) step(I1{}, t + 1);
  }
#ifdef VLA_Q4_STAMP
  const uint64_t st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- the second K part hands its sums to the first through LDS (the ring is free: every wave
  // has passed the last step's barrier with all its requests landed)
  {
    float* red = reinterpret_cast<float*>(lds) + wn * (NT * MT * 16 * 64) + lane;
    if (wk == 1) {
#pragma unroll
      for (int g = 0; g < NT; ++g)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int i = 0; i < 16; ++i) red[((g * MT + a) * 16 + i) * 64] = acc[g][a][i];
    }
    __syncthreads();
    if (wk == 1) return;
#pragma unroll
    for (int g = 0; g < NT; ++g)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][a][i] += red[((g * MT + a) * 16 + i) * 64];
  }
#pragma unroll
  for (int g = 0; g < NT; ++g)
    store_tile<SPLIT, MT>(acc[g], outp, m0 + c, m, wcol0 + 32 * g + 4 * h, n,
                          SPLIT ? (int64_t)blockIdx.z * m : (int64_t)0);
#ifdef VLA_Q4_STAMP
  {
    wait_vmcnt<0>();
    const uint64_t st_end = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && (wave == 0 || wave == 3) && blockIdx.z == 0 && ((blockIdx.x + 3 * blockIdx.y) % 37) == 0)
      printf("stamp2 wg %d,%d wave %d steps %d entry %llu loop0 +%llu loop1 +%llu end +%llu MHz %.0f cyc/step %.0f\n",
             blockIdx.x, blockIdx.y, wave, nsteps, (unsigned long long)(st_entry % 100000000ull),
             (unsigned long long)(st_r0 - st_entry), (unsigned long long)(st_r1 - st_entry),
             (unsigned long long)(st_end - st_entry),
             100.0 * (double)(st_c1 - st_c0) / (double)(st_r1 - st_r0), (double)(st_c1 - st_c0) / nsteps);
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// The LDS-DMA kernel for K-sliced layers: 128 x 32 wave tiles as in q4_gemm_dma_kernel, but the
// workgroup is 4 column waves x 2 K parts (8 waves, two per SIMD, a 128 x 128 output tile): the two
// parts' sums meet through LDS, so a layer that needs 256 workgroups' worth of K slices to fill the
// chip writes HALF as many fp32 slabs as with one K range per workgroup -- and the slabs are what
// these launches pay for: a 128 x 256 fp32 tile per workgroup leaves the chip at the HBM write
// rate (qkv_proj 4096 x 6144 at batch 256, four slices: 1.4 us to the first tile, 14.6 us of loop,
// 2.3 - 8 us until the last wave's 16 KiB x 8 are out), and the consumer reads them all back.
// Each K part has its own x ring (its 4 waves fill and read it); the barrier is the workgroup's.
__device__ __forceinline__ void glds_tile_kp(const uint32_t (&xoff)[4], uint64_t xbase, uint32_t xdst,
                                             uint32_t woff, uint64_t wbase, uint32_t wdst,
                                             const void* zsrc, uint32_t zdst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %10\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %10\n\t"
      "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %11\n\t"
      "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dword %8, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(xoff[0]), "v"(xoff[1]), "v"(xoff[2]), "v"(xoff[3]), "s"(xdst), "v"(woff), "s"(wdst),
        "v"(zsrc), "s"(zdst), "s"(xbase), "s"(wbase)
      : "memory");
}

template <int STAGES>
struct Q4DmaKp {
  static constexpr int XB = 128 * 128;          // x tile of one K part
  static constexpr int WB = 4 * 1024;           // 1 KiB of weight words per wave
  static constexpr int ZB = 4 * 256;            // 256 bytes of scales / zeros per wave
  static constexpr int PART = XB + WB + ZB;     // one K part's share of a stage
  static constexpr int STAGE = 2 * PART;
  static constexpr int BYTES = STAGES * STAGE;
  static constexpr int PER_TILE = 6;            // 4 x + 1 w + 1 zeros / scales
  static_assert(BYTES >= 4 * 128 * 32 * 4, "the K parts' sums meet in the ring's memory");
};

template <int ZOFF, bool SPLIT, int STAGES, int D>
__global__ __launch_bounds__(512) void q4_gemm_dma_kp_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split) {
  using L = Q4DmaKp<STAGES>;
  constexpr int MT = 4;
  static_assert(D >= 1 && STAGES >= D + 1 && STAGES <= 4 && L::PER_TILE * (D - 1) < 64, "ring depth");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // the only LDS object: offset 0

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave & 3, wk = wave >> 2;                 // column wave, K part
  const int h = lane >> 5, c = lane & 31;
  const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
  const int k_part = k_per_split / 2;                      // the host keeps this whole groups
  const int k_begin = blockIdx.z * k_per_split + wk * k_part;
  const int nsteps = k_part / TK;                          // the same for both parts
  if (nsteps <= 0) return;
  const int wcol0 = n0 + 32 * wn;                          // first column of this wave

  // ---- what this lane fetches (columns past n: clamped to column 0, never stored)
  uint32_t xoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wn * 4 + j) * 8 + (lane >> 3);
    const int q = (lane & 7) ^ ((r >> 1) & 7);
    xoff[j] = (uint32_t)(min(m0 + r, m - 1) * k + 8 * q) * 2u;       // M K < 2^30 (host check)
  }
  uint32_t wlane;
  {
    const int r = lane >> 3, col = wcol0 + 4 * (lane & 7);             // k8-row of the step, 4 columns
    wlane = (uint32_t)(r * n + (col < n ? col : 0)) * 4u;
  }
  const unsigned char* zsrc;                               // walks one group per `zs_stride`
  uint32_t zs_stride;
  if ((lane & 31) < 16 || (lane & 31) >= 20) {             // scale pairs (lanes 20.. repeat them)
    const int col = wcol0 + 2 * (lane & 15);
    zsrc = reinterpret_cast<const unsigned char*>(scales) + (size_t)(col < n ? col : 0) * 2;
    zs_stride = (uint32_t)n * 2u;
  } else {                                                 // lanes 16..19 of each half: zero words
    const int col = wcol0 + 8 * (lane & 3);
    zsrc = reinterpret_cast<const unsigned char*>(qz) + (size_t)((col < n ? col : 0) >> 3) * 4;
    zs_stride = (uint32_t)(n >> 3) * 4u;
  }
  zsrc += (size_t)(k_begin / group_size) * zs_stride;
  uint64_t xbase = reinterpret_cast<uint64_t>(x) + (uint64_t)k_begin * 2u;
  uint64_t wbase = reinterpret_cast<uint64_t>(qw) + (uint64_t)(k_begin >> 3) * (uint64_t)n * 4u;
  const uint64_t wstep = (uint64_t)(TK / 8) * (uint64_t)n * 4u;
  const int steps_per_group = group_size / TK;
  int group_left = steps_per_group - (k_begin / TK) % steps_per_group;
  const uint32_t part0 = (uint32_t)(wk * L::PART);
  const uint32_t xdst0 = part0 + (uint32_t)(wn * 4096), wdst0 = part0 + (uint32_t)(L::XB + wn * 1024),
                 zdst0 = part0 + (uint32_t)(L::XB + L::WB + wn * 256);
  auto issue = [&](int stage) {
    const uint32_t base = (uint32_t)(stage * L::STAGE);
    glds_tile_kp(xoff, xbase, base + xdst0, wlane, wbase, base + wdst0, zsrc, base + zdst0);
    xbase += TK * 2;
    wbase += wstep;
    if (--group_left == 0) {
      group_left = steps_per_group;
      zsrc += zs_stride;
    }
  };

  // ---- what this lane reads: column c of the wave, k 8h .. 8h + 7 of each 16-deep slice
  const int pbase = wk * L::PART;
  const int woff = pbase + L::XB + wn * 1024 + h * 128 + c * 4;     // + 256 s
  const int zsb = pbase + L::XB + L::WB + wn * 256;
  const int soff = zsb + 4 * (c >> 1), zoff = zsb + 64 + 4 * (c >> 3);
  const int zshift = 4 * (c & 7);
  const uint32_t ssel = (c & 1) ? 0x03020302u : 0x01000100u;
  const int aoff = pbase + c * 128;                        // + a 4096 + ((2 s + h) ^ cs) << 4
  const int cs = (c >> 1) & 7;

  f32x16 acc[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  uint32_t c1024 = 0x64006400u;
  asm volatile("" : "+v"(c1024));
  constexpr uint32_t kLoMask = 0x000f000fu, kHiMask = 0x00f000f0u;
  auto and_or = [](uint32_t a, uint32_t msk, uint32_t cc) { return (a & msk) | cc; };

  struct Words { uint32_t w[4], sw, zw; };                 // a tile's raw words of this lane
  struct Consts { f16x2 zc1, zc2, sc2; };
  auto load_words = [&](const unsigned char* st, Words& d) {
#pragma unroll
    for (int s = 0; s < 4; ++s) d.w[s] = *reinterpret_cast<const uint32_t*>(st + woff + 256 * s);
    d.sw = *reinterpret_cast<const uint32_t*>(st + soff);
    d.zw = *reinterpret_cast<const uint32_t*>(st + zoff);
  };
  auto make_consts = [&](const Words& d) {
    const uint32_t z = ((d.zw >> zshift) & 15u) + ZOFF;
    Consts r;
    r.zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);
    r.zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);
    r.sc2 = as_h2(__builtin_amdgcn_perm(d.sw, d.sw, ssel));
    return r;
  };
  auto read_a = [&](const unsigned char* st, int s, u32x4 (&af)[MT]) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
      af[a] = *reinterpret_cast<const u32x4*>(st + aoff + a * 4096 + (((2 * s + h) ^ cs) << 4));
  };
  auto dequant = [&](uint32_t wv, const Consts& q) {
    const uint32_t w8 = wv >> 8;
    const f16x2 d0 = as_h2(and_or(wv, kLoMask, c1024)) + q.zc1;
    const f16x2 d1 = __builtin_elementwise_fma(as_h2(and_or(wv, kHiMask, c1024)), sixteenth, q.zc2);
    const f16x2 d2 = as_h2(and_or(w8, kLoMask, c1024)) + q.zc1;
    const f16x2 d3 = __builtin_elementwise_fma(as_h2(and_or(w8, kHiMask, c1024)), sixteenth, q.zc2);
    u32x4 bf;
    bf[0] = as_u32(d0 * q.sc2);
    bf[1] = as_u32(d1 * q.sc2);
    bf[2] = as_u32(d2 * q.sc2);
    bf[3] = as_u32(d3 * q.sc2);
    return bf;
  };
  auto mma = [&](const u32x4 (&af)[MT], const u32x4& bf) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf),
                                                      __builtin_bit_cast(f16x8, af[a]), acc[a], 0, 0,
                                                      0);
  };
#define VLA_SB __builtin_amdgcn_sched_barrier(0)
  Words wd;                                                // words of the tile about to be computed
  auto step = [&](auto stage_c, int t) {
    constexpr int ST = decltype(stage_c)::value;
    const unsigned char* st = lds + ST * L::STAGE;
    const unsigned char* stn = lds + ((ST + 1) % STAGES) * L::STAGE;
    const Consts q = make_consts(wd);
    u32x4 fa[MT], fb[MT];
    read_a(st, 0, fa);
    u32x4 b0 = dequant(wd.w[0], q);
    VLA_SB;
    read_a(st, 1, fb);
    VLA_SB;
    mma(fa, b0);
    u32x4 b1 = dequant(wd.w[1], q);
    VLA_SB;
    if (t + D < nsteps) issue((ST + D) % STAGES);          // behind the first MFMAs, not before them
    VLA_SB;
    read_a(st, 2, fa);
    VLA_SB;
    mma(fb, b1);
    b0 = dequant(wd.w[2], q);
    VLA_SB;
    read_a(st, 3, fb);
    VLA_SB;
    mma(fa, b0);
    b1 = dequant(wd.w[3], q);
    VLA_SB;
    mma(fb, b1);
    VLA_SB;
    if (t + D < nsteps) wait_vmcnt<L::PER_TILE * (D - 1)>(); else wait_vmcnt<0>();
    if (t + 1 < nsteps) load_words(stn, wd);               // this wave's own bytes: no barrier needed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#undef VLA_SB

#ifdef VLA_Q4_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  // ---- prologue: tiles 0 .. D - 1 requested, tile 0 landed
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < nsteps) issue(j);
  if (D <= nsteps) wait_vmcnt<L::PER_TILE * (D - 1)>(); else wait_vmcnt<0>();
  load_words(lds, wd);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1 % STAGES>;
  using I2 = std::integral_constant<int, 2 % STAGES>;
  using I3 = std::integral_constant<int, 3 % STAGES>;
  {
    int t = 0;
    for (; t + STAGES <= nsteps; t += STAGES) {
      step(I0{}, t);
      step(I1{}, t + 1);
      if constexpr (STAGES >= 3) step(I2{}, t + 2);
      if constexpr (STAGES >= 4) step(I3{}, t + 3);
    }
    if (t < nsteps) step(I0{}, t);
    if (t + 1 < nsteps) step(I1{}, t + 1);
    if constexpr (STAGES >= 4) {
      if (t + 2 < nsteps) step(I2{}, t + 2);
    }
  }

  // ---- the second K part hands its sums to the first through LDS (the ring is free: every wave
  // has passed the last step's barrier with all its requests landed)
  {
    float* red = reinterpret_cast<float*>(lds) + wn * (MT * 16 * 64) + lane;
    if (wk == 1) {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(a * 16 + i) * 64] = acc[a][i];
    }
    __syncthreads();
    if (wk == 1) return;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][i] += red[(a * 16 + i) * 64];
  }
  store_tile<SPLIT, MT>(acc, outp, m0 + c, m, wcol0 + 4 * h, n,
                        SPLIT ? (int64_t)blockIdx.z * m : (int64_t)0);
}

// out[i] = fp16(sum over splits of slab[s][i]); 4 elements per thread
__global__ __launch_bounds__(256) void q4_splitk_reduce_kernel(const float* __restrict__ slabs,
                                                               _Float16* __restrict__ out,
                                                               int64_t elems, int splits) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= elems) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(slabs + i);
  for (int p = 1; p < splits; ++p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + (int64_t)p * elems + i);
    s += v;
  }
  u32x2 o;
  o[0] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)s[0]) |
         ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)s[1]) << 16);
  o[1] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)s[2]) |
         ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)s[3]) << 16);
  *reinterpret_cast<u32x2*>(out + i) = o;
}

// AWQ [K][N/8] (8 columns per word, nibble P[i] = column 8j+i, P = {0,4,1,5,2,6,3,7}) ->
// k-packed pair order [K/8][N]; one thread per output word
__global__ __launch_bounds__(256) void awq_repack_weight_kernel(const uint32_t* __restrict__ qweight,
                                                                uint32_t* __restrict__ out, int k,
                                                                int n) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int k8 = blockIdx.y;
  if (col >= n) return;
  const int i = col & 7, pos = 4 * ((i >> 1) + 4 * (i & 1));
  uint32_t w = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t q = (qweight[(int64_t)(k8 * 8 + e) * (n >> 3) + (col >> 3)] >> pos) & 15u;
    const int dst = (e & 1) ? 4 + (e >> 1) : (e >> 1);
    w |= q << (4 * dst);
  }
  out[(int64_t)k8 * n + col] = w;
}

// AWQ zeros [K/G][N/8] in AWQ nibble order -> natural nibble order (nibble i = column 8j+i)
__global__ __launch_bounds__(256) void awq_repack_zeros_kernel(const uint32_t* __restrict__ qzeros,
                                                               uint32_t* __restrict__ out,
                                                               int64_t words) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= words) return;
  const uint32_t z = qzeros[idx];
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= ((z >> (4 * ((i >> 1) + 4 * (i & 1)))) & 15u) << (4 * i);
  out[idx] = o;
}

}  // namespace
}  // namespace vla

using namespace vla;

namespace vla {
// q4_gemm_skinny.hip: batches of at most 16 rows
bool q4_skinny_shape(int m, int n, int k, int group, int* nwv, int* splits, int* cs);
int launch_q4_gemm_skinny(const void* x, const void* qw, const void* qz, const void* scales,
                          void* dst, int m, int n, int k, int group, int zero_offset, int nwv,
                          int splits, int cs, hipStream_t s);
bool q4_skinny_silu_ok(int m, int n, int k, int group);
int launch_q4_gemm_skinny_silu(const void* x, const void* qw, const void* qz, const void* scales,
                               void* out, int m, int n, int k, int group, int zero_offset,
                               hipStream_t s);
// q4_gemm_mid.hip: row blocks of 17..96 rows
bool q4_mid_shape(int m, int n, int k, int group, int* splits);
int launch_q4_gemm_mid(const void* x, const void* qw, const void* qz, const void* scales, void* dst,
                       int m, int n, int k, int group, int zero_offset, int splits, hipStream_t s);
int q4_mid_grouped_splits(int total_rows, int num_sets, int n, int k, int group);
int q4_mid_set_mode(int mode);
int launch_q4_gemm_mid_grouped(const void* x, const int* x_rows, const void* qw, const void* qz,
                               const void* scales, void* dst, const int* row_offsets, int num_sets,
                               int total_rows, int max_rows_per_set, int n, int k, int group,
                               int zero_offset, int splits, hipStream_t s);
bool q4_skinny_grouped_ok(int max_rows_per_set, int total_rows, int num_sets, int n, int k,
                          int group);
int launch_q4_gemm_skinny_grouped(const void* x, const int* x_rows, const void* qw, const void* qz,
                                  const void* scales, void* out, const int* row_offsets,
                                  int num_sets, int total_rows, int row_tiles, int n, int k,
                                  int group, int zero_offset, hipStream_t s);
}  // namespace vla

// The 32x32 kernel's launch shape: nw = waves (32 columns each) per workgroup, splits = K slices.
// From sweeps on MI355X at M = 256 over the Llama-3-8B layer shapes at TP 1/2/4/8:
//  * 256-column workgroups for wide, deep layers; 128-column ones (two fit a CU) when the layer is
//    narrow (N < 3072) or shallow (K <= 2048);
//  * as many K slices (at most 8) as keep the grid within one workgroup per CU -- a second round
//    of workgroups costs more than the emptier first one (N = 6144, K = 4096: 4 slices 31.6 us,
//    8 slices 39.8 us) -- but never slices shorter than 8 K steps: below that the fixed cost of
//    the slabs and their reduction exceeds what the shorter loop saves (K = 1024, N = 4096:
//    1 slice 15.3 us, 4 slices 19.2 us).
// VLA_Q4_TUNE="nw,splits[,mt]" overrides them (kernel studies).
struct Q4Shape { int nw, splits, mt, kw, kp; };   // kw: K parts inside a workgroup (small batches); kp: K parts of the 128-row LDS-DMA kernel (q4_gemm_dma_kp_kernel)

static Q4Shape pick_shape(int m, int n, int k, int group) {
  Q4Shape s;
  s.kw = 1;
  s.kp = 1;
  // Small batches (one or two 32-row tiles): the kernel streams the weights once, so the grid is
  // sized for bytes in flight -- 128-column workgroups and K slices until about one workgroup
  // per CU exists (the slabs of a 32-row batch are small) -- not for MFMA occupancy.
  s.mt = m <= 32 ? 1 : m <= 64 ? 2 : 4;
  const int mtiles = (m + 32 * s.mt - 1) / (32 * s.mt);
  int want;
  if (s.mt < 4) {
    // A 32- or 64-row step is a few hundred cycles of issue work per wave behind load, LDS and
    // MFMA-chain latencies, and the grid alone gives a SIMD about one wave: K parts inside the
    // workgroup add waves without adding fp32 slabs.  Measured at batch 8 / 32 / 64 (us,
    // "slices x parts"): gate_up 4096 x 28672 2x1 24.9 / 26.8 / 34.7 -> 1x4 22.3 / 23.7 / 31.0
    // (and no slabs at all); down 14336 x 4096 8x1 15.9 / 16.6 / 21.6 -> 8x2 14.5 / 15.3 / 19.9;
    // qkv 4096 x 6144 4x1 10.3 / 11.9 / 15.2 -> 4x2 9.8 / 11.3 / 14.9; o 4096 x 4096 8x1 9.5 /
    // 11.1 / 14.0 -> 4x2 9.5 / 10.5 / 13.2.  A part shorter than 8 K steps is all prologue.
    s.nw = 4;
    const int wgs = ((n + 127) / 128) * mtiles;
    if (wgs >= 192 && s.mt == 2) {
      // 64-row tiles of a wide layer: two slices of two K parts beat one slice of four
      // (gate_up 4096 x 28672 at 48 / 64 rows: 40.0 / 40.9 -> 36.6 / 38.3 us)
      want = 2;
      s.kw = 2;
    } else if (wgs >= 192) {
      want = 1;
      s.kw = 4;
    } else {
      s.kw = 2;
      want = (256 + wgs - 1) / wgs;
      if (want > 8) want = 8;
      if (want > k / (s.kw * 8 * TK)) want = k / (s.kw * 8 * TK);
    }
  } else if (m <= 256 && k < 4096 && ((n + 127) / 128) * ((m + 31) / 32) >= 192) {
    // Short contractions over many columns (the row-parallel layers of a tensor-parallel rank:
    // o_proj 512..2048 x 4096, down_proj 1792..3584 x 4096): a workgroup's loop is a few dozen
    // steps at most, so its fixed cost and the tail of the launch decide.  Smaller row tiles
    // fill the chip without slicing K, which also spares the slabs and their reduction:
    // 512 x 4096 9.6 -> 5.4 us, 1024 x 4096 17.3 -> 8.3, 2048 x 4096 22.6 -> 13.8,
    // 1792 x 4096 20.2 -> 12.2, 3584 x 4096 23.6 -> 21.5 (batch 256).  The largest tile height
    // that still gives 192 workgroups is taken.
    s.mt = ((n + 127) / 128) * ((m + 63) / 64) >= 192 ? 2 : 1;
    s.nw = 4;
    want = 1;
  } else if (m <= 256 && n <= 2048 && k >= 2048) {
    // Few columns, long contraction (qkv_proj of a tensor-parallel rank: 4096 x 768 / 1536):
    // four K slices of small row tiles instead of eight slices of 128-row tiles
    // (768 columns 16.5 -> 11.6 us, 1536 columns 19.5 -> 15.6).
    s.mt = n <= 1024 ? 1 : 2;
    s.nw = 4;
    want = 4;
  } else if (m <= 128 && n <= 4096 && k >= 4096) {
    // one 128-row tile would leave half of its rows or more empty and a layer of 4096 columns
    // with 16 or 32 workgroups per K slice: 64-row tiles, four slices (eight over a long K)
    // (batch 128: o_proj 4096 x 4096 23.6 -> 17.0 us, down_proj 14336 x 4096 36.3 -> 33.2)
    s.mt = 2;
    s.nw = 4;
    want = k > 8192 ? 8 : 4;
  } else if (m <= 256 && n < 4096 && k >= 4096 && k <= 8192) {
    // fewer than 4096 columns over a long contraction (gate_up of an 8-way rank, 4096 x 3584:
    // 24.3 -> 21.4 us)
    s.mt = 2;
    s.nw = 4;
    want = 4;
  } else {
    // a square 4096 x 4096 layer (o_proj): 128-column workgroups in four slices instead of
    // 256-column ones in eight -- the same 256 workgroups, half the fp32 slabs (25.4 -> 22.9 us
    // with the reduction launch, and whoever sums the slabs reads half as much)
    s.nw = (n < 3072 || k <= 2048 || (n <= 4096 && k <= 4096)) ? 4 : 8;
    const int wgs = ((n + 32 * s.nw - 1) / (32 * s.nw)) * mtiles;
    want = 256 / wgs;
    if (want > 8) want = 8;
    if (want > k / (8 * TK)) want = k / (8 * TK);
    if (const char* e = getenv("VLA_Q4_MAXSPLIT")) {   // kernel studies
      const int cap = atoi(e);
      if (cap >= 1 && want > cap) want = cap;
    }
    // K-sliced 128-row launches: 128-column workgroups of two K parts that meet in LDS
    // (q4_gemm_dma_kp_kernel) -- the same number of workgroups and of K steps per wave from half
    // the slices, i.e. half the fp32 slabs written here and read back by the consumer.  Parts
    // shorter than 8 K steps are all prologue.  The price is in the requests: a wave of a K part
    // issues four x pieces per K step where a wave of a 256-column workgroup issues two (each part
    // stages its own x tile).  Kernel traces of the decode step at batch 256 (kernel / the consumer
    // that sums its slabs, us): qkv_proj 4096 x 6144 22.3 / 10.7 -> 20.8 / 9.2, o_proj 4096 x 4096
    // 16.7 / 7.8 -> ~15 / 6.4, down_proj 14336 x 4096 35.1 / 7.8 -> 33.7 / 6.4; everything of a
    // layer but paged attention 214.1 -> 207.2.  VLA_Q4_KP=0 switches the K parts off.
    static const bool kp_on = !(getenv("VLA_Q4_KP") && atoi(getenv("VLA_Q4_KP")) == 0);
    if (kp_on && want > 1 && group % TK == 0) {
      const int wgs_kp = ((n + 127) / 128) * mtiles;
      int want_kp = 256 / wgs_kp;
      if (want_kp > 8) want_kp = 8;
      const int unit2 = 2 * (group > TK ? group : TK);
      while (want_kp > 1 && (k / (want_kp * 2 * TK) < 8 || k % (want_kp * unit2) != 0)) --want_kp;
      if (want_kp >= 1 && k % (want_kp * unit2) == 0 && k / (want_kp * 2 * TK) >= 8 &&
          wgs_kp * want_kp >= 128) {
        s.nw = 4;
        s.kp = 2;
        want = want_kp;
      }
    }
  }
  if (want < 1) want = 1;
  int kw = s.kw;
  if (const char* e = getenv("VLA_Q4_TUNE")) {
    int a = 0, b = 0, c = 0, d = 0;
    const int got = sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &d);
    if (got >= 3 && (c == 1 || c == 2 || c == 4)) s.mt = c;   // "nw,splits,mt[,kw]"
    if (got >= 2 && (a == 4 || (a == 8 && s.mt == 4)) && b >= 1 && b <= 16) {
      s.nw = a;
      want = b;
    }
    if (got == 4 && (d == 1 || d == 2 || d == 4)) kw = d;
    // 32- and 64-row tiles exist with four waves (128 columns) only: an override that shrinks
    // the tile must not leave the 256-column grid of the shape it replaced
    if (s.mt != 4) s.nw = 4;
    s.kp = 1;
    if (const char* e2 = getenv("VLA_Q4_TUNE_KP")) {    // with "4,splits" above: the K-part kernel
      if (atoi(e2) == 2 && s.mt == 4 && s.nw == 4) s.kp = 2;
    }
  }
  const int unit = (group > TK ? group : TK) * s.kp;  // a split must hold whole groups and whole K steps (per K part)
  while (want > 1 && (k % (want * unit)) != 0) --want;
  if (s.kp == 2 && k % (want * unit) != 0) s.kp = 1;
  s.splits = want;
  // K parts inside the workgroup: small row tiles of 128-column workgroups only, whole groups
  // (pairs of K steps where the per-group constants are fetched per pair) in every part
  if (s.mt == 4 || s.nw != 4) kw = 1;
  while (kw > 1 && ((k / want) % (kw * (unit > 2 * TK ? unit : 2 * TK))) != 0) kw >>= 1;
  s.kw = kw;
  return s;
}

// The LDS-DMA kernel's ring is dynamic LDS beyond the 64 KiB a kernel gets by default: the limit
// is raised once per device and instantiation.
template <int Z, bool SP, int NW, int ST, int D>
static hipError_t launch_dma(dim3 grid, hipStream_t s, const _Float16* x, const uint32_t* qw,
                             const uint32_t* qz, const _Float16* sc, void* dst, int m, int n, int k,
                             int group, int kps) {
  using L = Q4Dma<NW, ST>;
  auto fn = q4_gemm_dma_kernel<Z, SP, NW, ST, D>;
  static std::atomic<uint64_t> ready{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (!((ready.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                            hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
    if (e != hipSuccess) return e;
    ready.fetch_or(1ull << (dev & 63), std::memory_order_release);
  }
  hipLaunchKernelGGL(fn, grid, dim3(64 * NW), L::BYTES, s, x, qw, qz, sc, dst, m, n, k, group, kps);
  return hipSuccess;
}

template <int Z, bool SP, int ST, int D>
static hipError_t launch_dma2(dim3 grid, hipStream_t s, const _Float16* x, const uint32_t* qw,
                              const uint32_t* qz, const _Float16* sc, void* dst, int m, int n, int k,
                              int group, int kps) {
  using L = Q4Dma2<ST>;
  auto fn = q4_gemm_dma2_kernel<Z, SP, ST, D>;
  static std::atomic<uint64_t> ready{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (!((ready.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                            hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
    if (e != hipSuccess) return e;
    ready.fetch_or(1ull << (dev & 63), std::memory_order_release);
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), L::BYTES, s, x, qw, qz, sc, dst, m, n, k, group, kps);
  return hipSuccess;
}

template <int Z, bool SP, int ST, int D>
static hipError_t launch_dma_kp(dim3 grid, hipStream_t s, const _Float16* x, const uint32_t* qw,
                                const uint32_t* qz, const _Float16* sc, void* dst, int m, int n, int k,
                                int group, int kps) {
  using L = Q4DmaKp<ST>;
  auto fn = q4_gemm_dma_kp_kernel<Z, SP, ST, D>;
  static std::atomic<uint64_t> ready{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (!((ready.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                            hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
    if (e != hipSuccess) return e;
    ready.fetch_or(1ull << (dev & 63), std::memory_order_release);
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), L::BYTES, s, x, qw, qz, sc, dst, m, n, k, group, kps);
  return hipSuccess;
}

extern "C" int vla_set_q4_row_block_kernel(int mode) {
  return q4_mid_set_mode(mode) ? VLA_ERR_ARG : VLA_OK;
}

extern "C" int64_t vla_q4_gemm_workspace_bytes(int m, int n, int k, int group) {
  if (m <= 0 || n <= 0 || k <= 0 || group <= 0) return 0;
  int sk_nwv = 0, sk_splits = 0, sk_cs = 0;
  int md_splits = 0;
  const int s = q4_skinny_shape(m, n, k, group, &sk_nwv, &sk_splits, &sk_cs)
                    ? sk_splits
                    : q4_mid_shape(m, n, k, group, &md_splits) ? md_splits
                                                               : pick_shape(m, n, k, group).splits;
  return s > 1 ? (int64_t)s * m * n * 4 : 0;
}

// reduce: sum the split-K slabs into `out` here; otherwise leave them in the workspace and report
// their number (0: the product went straight to `out`) for a consumer that sums them itself
static int q4_gemm_run(const void* x, const int32_t* qweight, const int32_t* qzeros,
                       const void* scales, void* out, int m, int n, int k, int group,
                       int zero_offset, void* workspace, int64_t workspace_bytes, void* stream,
                       bool reduce, int* num_slabs) {
  (void)hipGetLastError();
  if (num_slabs) *num_slabs = 0;
  if (!x || !qweight || !qzeros || !scales || !out || m < 0) return VLA_ERR_ARG;
  if (n % 8 || k % TK || group <= 0 || group % 32 || k % group ||
      (zero_offset != 0 && zero_offset != 1))
    return VLA_ERR_QUANT_SHAPE;
  if (((uintptr_t)x % 16) || ((uintptr_t)out % 8)) return VLA_ERR_ARG;
  if (m == 0) return VLA_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool m32 = group % TK == 0;                  // the 32x32 kernel wants whole K steps per group
  static const bool g2_enabled = !(getenv("VLA_Q4_G2") && atoi(getenv("VLA_Q4_G2")) == 0);  // kernel studies
  // 128-row tiles: the LDS-DMA kernels.  VLA_Q4_DMA (kernel studies): 0 the register-staged kernel,
  // 3 / 4 the 128 x 32 wave tile over that many ring stages, 23 the 128 x 64 wave tile wherever it
  // fits; unset: 128 x 64 wave tiles for unsliced layers whose K parts are at least 16 steps
  // (gate_up 4096 x 28672 at batch 256: 74 -> 69 us), 128 x 32 tiles over 3 stages otherwise
  // (qkv / o / down: 30.3 / 25.2 / 45.5 -> 29.3 / 24.6 / 44.3 us).
  static const int dma_env = (getenv("VLA_Q4_DMA") && *getenv("VLA_Q4_DMA")) ? atoi(getenv("VLA_Q4_DMA")) : -1;
  const bool dma_ok = (int64_t)m * k < (1ll << 30) && (int64_t)n * 32 < (1ll << 30);   // 32-bit lane offsets
  int sk_nwv = 0, sk_splits = 0, sk_cs = 0;
  if (q4_skinny_shape(m, n, k, group, &sk_nwv, &sk_splits, &sk_cs)) {
    // at most 16 rows: one pass over the weights, see q4_gemm_skinny.hip
    if (sk_splits > 1 && (!workspace || workspace_bytes < (int64_t)sk_splits * m * n * 4 ||
                          ((uintptr_t)workspace % 16) || ((int64_t)m * n) % 4))
      sk_splits = 1;
    launch_q4_gemm_skinny(x, qweight, qzeros, scales, sk_splits > 1 ? workspace : out, m, n, k,
                          group, zero_offset, sk_nwv, sk_splits, sk_cs, s);
    VLA_HIP_CHECK(hipGetLastError());
    if (sk_splits > 1 && reduce) {
      const int64_t elems = (int64_t)m * n;
      hipLaunchKernelGGL(q4_splitk_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)),
                         dim3(256), 0, s, (const float*)workspace, (_Float16*)out, elems, sk_splits);
      VLA_HIP_CHECK(hipGetLastError());
    } else if (sk_splits > 1 && num_slabs) {
      *num_slabs = sk_splits;
    }
    return VLA_OK;
  }
  int md_splits = 0;
  if (q4_mid_shape(m, n, k, group, &md_splits)) {
    // 17..96 rows: one pass over the weights with x shared through LDS, see q4_gemm_mid.hip
    if (md_splits > 1 && (!workspace || workspace_bytes < (int64_t)md_splits * m * n * 4 ||
                          ((uintptr_t)workspace % 16) || ((int64_t)m * n) % 4))
      md_splits = 1;
    launch_q4_gemm_mid(x, qweight, qzeros, scales, md_splits > 1 ? workspace : out, m, n, k, group,
                       zero_offset, md_splits, s);
    VLA_HIP_CHECK(hipGetLastError());
    if (md_splits > 1 && reduce) {
      const int64_t elems = (int64_t)m * n;
      hipLaunchKernelGGL(q4_splitk_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)),
                         dim3(256), 0, s, (const float*)workspace, (_Float16*)out, elems, md_splits);
      VLA_HIP_CHECK(hipGetLastError());
    } else if (md_splits > 1 && num_slabs) {
      *num_slabs = md_splits;
    }
    return VLA_OK;
  }
  Q4Shape shape = pick_shape(m, n, k, group);
  int splits = shape.splits;
  int dma_stages = dma_env;            // decided below, once the slicing is final
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * m * n * 4 ||
                     ((uintptr_t)workspace % 16) || ((int64_t)m * n) % 4))
    splits = 1;
  const int tn = m32 ? 32 * shape.nw : TN, tm = m32 ? 32 * shape.mt : TM;
  dim3 grid((n + tn - 1) / tn, (m + tm - 1) / tm, splits);
  const int kps = k / splits;
  void* dst = splits > 1 ? workspace : out;
  const bool parts_ok = kps % (2 * (group > TK ? group : TK)) == 0;   // two K parts of whole groups
  if (dma_env < 0) dma_stages = (splits == 1 && shape.nw == 8 && parts_ok && kps / (2 * TK) >= 16) ? 23 : 3;
  if (dma_stages == 23 && !(shape.nw == 8 && parts_ok)) dma_stages = 3;
#define VLA_Q4_ARGS                                                                            \
  (const _Float16*)x, (const uint32_t*)qweight, (const uint32_t*)qzeros, (const _Float16*)scales, \
      dst, m, n, k, group, kps
#define VLA_Q4_LAUNCH_KW(Z, SP, G2, MT_)                                                       \
  do {                                                                                         \
    if (shape.kw == 4)                                                                         \
      hipLaunchKernelGGL((q4_gemm_m32_kernel<Z, SP, 4, MT_, false, G2, 4>), grid, dim3(1024),  \
                         0, s, VLA_Q4_ARGS);                                                   \
    else if (shape.kw == 2)                                                                    \
      hipLaunchKernelGGL((q4_gemm_m32_kernel<Z, SP, 4, MT_, false, G2, 2>), grid, dim3(512),   \
                         0, s, VLA_Q4_ARGS);                                                   \
    else                                                                                       \
      hipLaunchKernelGGL((q4_gemm_m32_kernel<Z, SP, 4, MT_, false, G2, 1>), grid, dim3(256),   \
                         0, s, VLA_Q4_ARGS);                                                   \
  } while (0)
#define VLA_Q4_LAUNCH_G(Z, SP, G2)                                                             \
  do {                                                                                         \
    if (shape.mt == 1)                                                                         \
      VLA_Q4_LAUNCH_KW(Z, SP, G2, 1);                                                          \
    else if (shape.mt == 2)                                                                    \
      VLA_Q4_LAUNCH_KW(Z, SP, G2, 2);                                                          \
    else if (dma_ok && shape.kp == 2 && shape.nw == 4 && parts_ok && dma_env != 0)             \
      VLA_HIP_CHECK((launch_dma_kp<Z, SP, 3, 2>(grid, s, VLA_Q4_ARGS)));                       \
    else if (dma_ok && dma_stages == 23 && shape.nw == 8 && parts_ok)                          \
      VLA_HIP_CHECK((launch_dma2<Z, SP, 3, 2>(grid, s, VLA_Q4_ARGS)));                         \
    else if (dma_ok && dma_stages == 3 && shape.nw == 8)                                       \
      VLA_HIP_CHECK((launch_dma<Z, SP, 8, 3, 2>(grid, s, VLA_Q4_ARGS)));                       \
    else if (dma_ok && dma_stages == 3 && shape.nw == 4)                                       \
      VLA_HIP_CHECK((launch_dma<Z, SP, 4, 3, 2>(grid, s, VLA_Q4_ARGS)));                       \
    else if (dma_ok && dma_stages == 4 && shape.nw == 8)                                       \
      VLA_HIP_CHECK((launch_dma<Z, SP, 8, 4, 3>(grid, s, VLA_Q4_ARGS)));                       \
    else if (dma_ok && dma_stages == 4 && shape.nw == 4)                                       \
      VLA_HIP_CHECK((launch_dma<Z, SP, 4, 4, 3>(grid, s, VLA_Q4_ARGS)));                       \
    else if (shape.nw == 8)                                                                    \
      hipLaunchKernelGGL((q4_gemm_m32_kernel<Z, SP, 8, 4, false, G2>), grid, dim3(512), 0, s,  \
                         VLA_Q4_ARGS);                                                         \
    else                                                                                       \
      hipLaunchKernelGGL((q4_gemm_m32_kernel<Z, SP, 4, 4, false, G2>), grid, dim3(256), 0, s,  \
                         VLA_Q4_ARGS);                                                         \
  } while (0)
#define VLA_Q4_LAUNCH(Z, SP)                                                                   \
  do {                                                                                         \
    if (!m32)                                                                                  \
      hipLaunchKernelGGL((q4_gemm_fast_kernel<Z, SP>), grid, dim3(kThreadsG), 0, s,            \
                         (const _Float16*)x, (const uint32_t*)qweight, (const uint32_t*)qzeros, \
                         (const _Float16*)scales, dst, m, n, k, group, kps);                   \
    else if (group == 2 * TK && g2_enabled)                                                    \
      VLA_Q4_LAUNCH_G(Z, SP, true);                                                            \
    else                                                                                       \
      VLA_Q4_LAUNCH_G(Z, SP, false);                                                           \
  } while (0)
  if (zero_offset == 0) {
    if (splits > 1) VLA_Q4_LAUNCH(0, true); else VLA_Q4_LAUNCH(0, false);
  } else {
    if (splits > 1) VLA_Q4_LAUNCH(1, true); else VLA_Q4_LAUNCH(1, false);
  }
#undef VLA_Q4_LAUNCH
#undef VLA_Q4_LAUNCH_G
#undef VLA_Q4_LAUNCH_KW
#undef VLA_Q4_ARGS
  VLA_HIP_CHECK(hipGetLastError());
  if (splits > 1 && reduce) {
    const int64_t elems = (int64_t)m * n;
    hipLaunchKernelGGL(q4_splitk_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)),
                       dim3(256), 0, s, (const float*)workspace, (_Float16*)out, elems, splits);
    VLA_HIP_CHECK(hipGetLastError());
  } else if (splits > 1 && num_slabs) {
    *num_slabs = splits;
  }
  return VLA_OK;
}

extern "C" int vla_q4_gemm_packed(const void* x, const int32_t* qweight, const int32_t* qzeros,
                                  const void* scales, void* out, int m, int n, int k, int group,
                                  int zero_offset, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  return q4_gemm_run(x, qweight, qzeros, scales, out, m, n, k, group, zero_offset, workspace,
                     workspace_bytes, stream, true, nullptr);
}

extern "C" int vla_q4_gemm_packed_partial(const void* x, const int32_t* qweight,
                                          const int32_t* qzeros, const void* scales, void* out,
                                          int m, int n, int k, int group, int zero_offset,
                                          void* workspace, int64_t workspace_bytes,
                                          int* num_slabs, void* stream) {
  if (!num_slabs) return VLA_ERR_ARG;
  return q4_gemm_run(x, qweight, qzeros, scales, out, m, n, k, group, zero_offset, workspace,
                     workspace_bytes, stream, false, num_slabs);
}

// x [m, k] . [gate | up] (k-packed, n columns) -> silu(gate) * up [m, n / 2] in one launch, for
// the shapes the small-batch kernel runs unsliced (vla_q4_gemm_silu_and_mul_supported; the
// caller otherwise runs the product and the activation as two launches, which give the same bits).
extern "C" int vla_q4_gemm_silu_and_mul_supported(int m, int n, int k, int group) {
  if (m <= 0 || n <= 0 || k <= 0 || group <= 0 || k % TK || group % 32 || k % group) return 0;
  return q4_skinny_silu_ok(m, n, k, group) ? 1 : 0;
}

extern "C" int vla_q4_gemm_packed_silu_and_mul(const void* x, const int32_t* qweight,
                                               const int32_t* qzeros, const void* scales,
                                               void* out, int m, int n, int k, int group,
                                               int zero_offset, void* stream) {
  (void)hipGetLastError();
  if (!x || !qweight || !qzeros || !scales || !out || m < 0) return VLA_ERR_ARG;
  if (n % 16 || k % TK || group <= 0 || group % 32 || k % group ||
      (zero_offset != 0 && zero_offset != 1))
    return VLA_ERR_QUANT_SHAPE;
  if (((uintptr_t)x % 16) || ((uintptr_t)out % 8)) return VLA_ERR_ARG;
  if (m == 0) return VLA_OK;
  if (!q4_skinny_silu_ok(m, n, k, group)) return VLA_ERR_QUANT_SHAPE;
  launch_q4_gemm_skinny_silu(x, qweight, qzeros, scales, out, m, n, k, group, zero_offset,
                             (hipStream_t)stream);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_awq_repack(const int32_t* qweight, const int32_t* qzeros, int32_t* out_qweight,
                              int32_t* out_qzeros, int in_c, int out_c, int group, void* stream) {
  (void)hipGetLastError();
  if (!qweight || !qzeros || !out_qweight || !out_qzeros) return VLA_ERR_ARG;
  if (in_c <= 0 || out_c <= 0 || in_c % 8 || out_c % 8 || group <= 0 || in_c % group)
    return VLA_ERR_QUANT_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(awq_repack_weight_kernel, dim3((out_c + 255) / 256, in_c / 8), dim3(256), 0, s,
                     (const uint32_t*)qweight, (uint32_t*)out_qweight, in_c, out_c);
  const int64_t zw = (int64_t)(in_c / group) * (out_c / 8);
  hipLaunchKernelGGL(awq_repack_zeros_kernel, dim3((unsigned)((zw + 255) / 256)), dim3(256), 0, s,
                     (const uint32_t*)qzeros, (uint32_t*)out_qzeros, zw);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

// Grouped product for small row blocks (MoE experts at decode): 64-row tiles, 128-column
// workgroups, K split into at most 8 slices until the launch has about two workgroups per CU per
// expected active tile; the slabs are summed by the reduction kernel.
static int q4_grouped_tile_splits(int total_rows, int n, int k, int group, int num_sets) {
  const int wgs = ((n + 127) / 128) * num_sets;
  int want = (512 + wgs - 1) / wgs;
  if (want > 8) want = 8;
  if (want > k / (4 * TK)) want = k / (4 * TK);
  if (want < 1) want = 1;
  const int unit = group > TK ? group : TK;
  while (want > 1 && (k % (want * unit)) != 0) --want;
  (void)total_rows;
  return want;
}

// the larger of what the two grouped kernels (128-row tiles, 80-row tiles of 16-row blocks) would
// use for this product; each takes its own slice count out of it
extern "C" int64_t vla_q4_gemm_grouped_workspace_bytes(int total_rows, int n, int k, int group,
                                                       int num_sets) {
  if (total_rows <= 0 || n <= 0 || k <= 0 || group <= 0 || num_sets <= 0) return 0;
  int want = q4_grouped_tile_splits(total_rows, n, k, group, num_sets);
  const int mid = q4_mid_grouped_splits(total_rows, num_sets, n, k, group);
  if (mid > want) want = mid;
  return want > 1 ? (int64_t)want * total_rows * n * 4 : 0;
}

// x_rows: null, or the row of x each product row reads (the gather of the routed tokens folded
// into the tile staging).  clear_slabs: rows outside every set come back zero from a K-split
// launch; a caller that never reads such rows can spare the memset.
static int q4_gemm_grouped_run(const void* x, const int* x_rows, const int32_t* qweight,
                               const int32_t* qzeros, const void* scales, void* out,
                               const int* row_offsets, int num_sets, int total_rows,
                               int max_rows_per_set, int n, int k, int group, int zero_offset,
                               void* workspace, int64_t workspace_bytes, bool clear_slabs,
                               void* stream) {
  (void)hipGetLastError();
  if (!x || !qweight || !qzeros || !scales || !out || !row_offsets || num_sets <= 0 ||
      total_rows < 0 || max_rows_per_set < 0)
    return VLA_ERR_ARG;
  if (n % 8 || k % TK || group <= 0 || group % TK || k % group ||
      (zero_offset != 0 && zero_offset != 1))
    return VLA_ERR_QUANT_SHAPE;
  if (((uintptr_t)x % 16) || ((uintptr_t)out % 8)) return VLA_ERR_ARG;
  if (total_rows == 0 || max_rows_per_set == 0) return VLA_OK;
  if (q4_skinny_grouped_ok(max_rows_per_set, total_rows, num_sets, n, k, group)) {
    // a few rows per set (a decode batch of a few tokens): the LDS-free small-batch kernel, one
    // launch over all sets and their 16-row tiles, no K slices
    launch_q4_gemm_skinny_grouped(x, x_rows, qweight, qzeros, scales, out, row_offsets, num_sets,
                                  total_rows, (max_rows_per_set + 15) / 16, n, k, group,
                                  zero_offset, (hipStream_t)stream);
    VLA_HIP_CHECK(hipGetLastError());
    return VLA_OK;
  }
  const bool ws_ok = workspace && ((uintptr_t)workspace % 16) == 0 && ((int64_t)total_rows * n) % 4 == 0;
  if (int md = q4_mid_grouped_splits(total_rows, num_sets, n, k, group)) {
    // a few dozen rows per set: 80-row tiles of 16-row blocks with x shared through LDS, one
    // launch over all sets; K slices for narrow products
    const int64_t md_need = (int64_t)md * total_rows * n * 4;
    if (md == 1 || (ws_ok && workspace_bytes >= md_need)) {
      hipStream_t ms = (hipStream_t)stream;
      if (md > 1 && clear_slabs) VLA_HIP_CHECK(hipMemsetAsync(workspace, 0, (size_t)md_need, ms));
      launch_q4_gemm_mid_grouped(x, x_rows, qweight, qzeros, scales, md > 1 ? workspace : out,
                                 row_offsets, num_sets, total_rows, max_rows_per_set, n, k, group,
                                 zero_offset, md, ms);
      VLA_HIP_CHECK(hipGetLastError());
      if (md > 1) {
        const int64_t elems = (int64_t)total_rows * n;
        hipLaunchKernelGGL(q4_splitk_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)),
                           dim3(256), 0, ms, (const float*)workspace, (_Float16*)out, elems, md);
        VLA_HIP_CHECK(hipGetLastError());
      }
      return VLA_OK;
    }
  }
  int splits = q4_grouped_tile_splits(total_rows, n, k, group, num_sets);
  const int64_t need = splits > 1 ? (int64_t)splits * total_rows * n * 4 : 0;
  if (splits > 1 && (!ws_ok || workspace_bytes < need)) splits = 1;
  Q4Groups gp{};
  gp.row_offsets = row_offsets;
  gp.x_rows = x_rows;
  // Tile height.  A workgroup runs the body built for its tile's live 32-row sub-tiles, so a
  // tall tile costs a short row block registers, not MFMAs, while a short tile costs a row block
  // that overflows it a second pass over the set's weights -- and routing spreads rows unevenly
  // (top-2 of 8 over 256 tokens: 64 rows per expert on average, half of the experts above that).
  // Mixtral-8x7B block from a graph, 64- against 128-row tile templates on one box: 16 tokens
  // 0.236 / 0.212-0.223 ms, 64 tokens 0.237 / 0.223-0.225, 96 tokens 0.242-0.256 / 0.227-0.231,
  // 256 tokens 0.409 / 0.377, 1024 tokens 1.04 / 0.98: 128-row tiles at every size.
  // VLA_Q4_GROUPED_MT=2/4 overrides (kernel studies).
  int mt = 4;
  if (const char* e = getenv("VLA_Q4_GROUPED_MT")) {
    const int v = atoi(e);
    if (v == 2 || v == 4) mt = v;
  }
  gp.tiles_per_set = (max_rows_per_set + 32 * mt - 1) / (32 * mt);
  gp.total_rows = total_rows;
  gp.w_stride = (int64_t)(k / 8) * n;
  gp.z_stride = (int64_t)(k / group) * (n / 8);
  gp.s_stride = (int64_t)(k / group) * n;
  hipStream_t s = (hipStream_t)stream;
  // 256-column workgroups (8 waves, one workgroup per CU) for wide layers whose sets fill their
  // 128-row tiles: Mixtral-8x7B block at 1024 tokens (256 rows per expert on average) 1.00 ->
  // 0.94 ms; at 64-512 tokens (16-128 rows per expert) 128-column workgroups, two per CU, stay
  // ahead or level.  VLA_Q4_GROUPED_NW=4 / 8 forces either (kernel studies).
  static const int grouped_nw = getenv("VLA_Q4_GROUPED_NW") ? atoi(getenv("VLA_Q4_GROUPED_NW")) : 0;
  const bool nw8 = mt == 4 && n >= 8192 && splits == 1 &&
                   (grouped_nw == 8 || (grouped_nw == 0 && total_rows >= 192 * num_sets));
  dim3 grid(nw8 ? (n + 255) / 256 : (n + 127) / 128, num_sets * gp.tiles_per_set, splits);
  const int kps = k / splits;
  // sets of a few rows (an MoE block at 16-64 decode tokens): two K parts per workgroup (see
  // pick_shape), whole groups and at least eight K steps in each.  Mixtral-8x7B block from a
  // graph: 64 tokens 0.412 -> 0.386 ms; at 256 tokens and more the tiles are full enough that
  // the second part only costs (0.495 -> 0.546 ms), so it stays off there.
  // VLA_Q4_GROUPED_KW=1 switches it off (kernel studies)
  const int part_unit = group > 2 * TK ? group : 2 * TK;
  static const bool kw_enabled = !(getenv("VLA_Q4_GROUPED_KW") && atoi(getenv("VLA_Q4_GROUPED_KW")) == 1);
  const int kw = (kw_enabled && mt == 2 && total_rows <= 32 * num_sets &&
                  kps % (2 * part_unit) == 0 && kps / (2 * TK) >= 8) ? 2 : 1;
  void* dst = splits > 1 ? workspace : out;
  if (splits > 1 && clear_slabs) {
    // rows that belong to no set are never written: their sums must not be junk
    VLA_HIP_CHECK(hipMemsetAsync(workspace, 0, (size_t)need, s));
  }
#define VLA_Q4G_ARGS                                                                            \
  (const _Float16*)x, (const uint32_t*)qweight, (const uint32_t*)qzeros, (const _Float16*)scales, \
      dst, 0, n, k, group, kps, gp
#define VLA_Q4G_G(Z, SP, G2)                                                                    \
  do {                                                                                          \
    if (mt == 4 && nw8)                                                                         \
      hipLaunchKernelGGL((q4_gemm_m32_kernel<Z, SP, 8, 4, true, G2>), grid, dim3(512), 0, s,    \
                         VLA_Q4G_ARGS);                                                         \
    else if (mt == 4)                                                                           \
      hipLaunchKernelGGL((q4_gemm_m32_kernel<Z, SP, 4, 4, true, G2>), grid, dim3(256), 0, s,    \
                         VLA_Q4G_ARGS);                                                         \
    else if (kw == 2)                                                                           \
      hipLaunchKernelGGL((q4_gemm_m32_kernel<Z, SP, 4, 2, true, G2, 2>), grid, dim3(512), 0, s, \
                         VLA_Q4G_ARGS);                                                         \
    else                                                                                        \
      hipLaunchKernelGGL((q4_gemm_m32_kernel<Z, SP, 4, 2, true, G2>), grid, dim3(256), 0, s,    \
                         VLA_Q4G_ARGS);                                                         \
  } while (0)
#define VLA_Q4G(Z, SP)                                                                          \
  do {                                                                                          \
    if (group == 2 * TK) VLA_Q4G_G(Z, SP, true); else VLA_Q4G_G(Z, SP, false);                  \
  } while (0)
  if (zero_offset == 0) {
    if (splits > 1) VLA_Q4G(0, true); else VLA_Q4G(0, false);
  } else {
    if (splits > 1) VLA_Q4G(1, true); else VLA_Q4G(1, false);
  }
#undef VLA_Q4G
#undef VLA_Q4G_G
#undef VLA_Q4G_ARGS
  VLA_HIP_CHECK(hipGetLastError());
  if (splits > 1) {
    const int64_t elems = (int64_t)total_rows * n;
    hipLaunchKernelGGL(q4_splitk_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)),
                       dim3(256), 0, s, (const float*)workspace, (_Float16*)out, elems, splits);
    VLA_HIP_CHECK(hipGetLastError());
  }
  return VLA_OK;
}

extern "C" int vla_q4_gemm_grouped(const void* x, const int32_t* qweight, const int32_t* qzeros,
                                   const void* scales, void* out, const int* row_offsets,
                                   int num_sets, int total_rows, int max_rows_per_set, int n, int k,
                                   int group, int zero_offset, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  return q4_gemm_grouped_run(x, nullptr, qweight, qzeros, scales, out, row_offsets, num_sets,
                             total_rows, max_rows_per_set, n, k, group, zero_offset, workspace,
                             workspace_bytes, true, stream);
}

extern "C" int vla_q4_gemm_grouped_gather(const void* x, const int* x_rows, const int32_t* qweight,
                                          const int32_t* qzeros, const void* scales, void* out,
                                          const int* row_offsets, int num_sets, int total_rows,
                                          int max_rows_per_set, int n, int k, int group,
                                          int zero_offset, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  return q4_gemm_grouped_run(x, x_rows, qweight, qzeros, scales, out, row_offsets, num_sets,
                             total_rows, max_rows_per_set, n, k, group, zero_offset, workspace,
                             workspace_bytes, false, stream);
}

