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

#include "vla_common.h"

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

// GROUPED (several weight sets in one launch, e.g. the experts of an MoE block): x and out hold
// the row blocks of all sets back to back, gp.row_offsets[e] .. [e + 1] on the device says which
// rows belong to set e, blockIdx.y = e * tiles_per_set + row tile, and a workgroup whose tile
// lies past its set's rows leaves at once.
struct Q4Groups {
  const int* row_offsets;      // [num_sets + 1]
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
template <int ZOFF, bool SPLIT, int NW, int MT, bool GROUPED = false, bool G2 = false, int KW = 1>
__global__ __launch_bounds__(64 * NW * KW) void q4_gemm_m32_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split, Q4Groups gp = Q4Groups{}) {
  constexpr int THREADS = 64 * NW;               // threads of one K part
  constexpr int TMW = 32 * MT;                   // rows per workgroup
  constexpr int X2_TILE_BYTES = TMW * TK * 2;    // one LDS buffer: 16 KiB at MT = 4
  static_assert((TMW * 8) % THREADS == 0, "x staging covers the tile in whole passes");
  static_assert(KW == 1 || (NW <= 4 && (KW == 2 || KW == 4)), "LDS reduction fits the x buffers");
  constexpr int XCH = (TMW * 8) / THREADS;  // 16-byte chunks of the x tile per thread
  constexpr int XROWS = THREADS / 8;        // rows covered by one pass of the staging threads
  __shared__ __attribute__((aligned(16))) unsigned char xs_all[KW * 2 * X2_TILE_BYTES];

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
    x += (int64_t)row_base * k;
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
  auto phys = [&](int t) { return min(t, nsteps - 1); };   // past the end: re-read the last step

  // ---- x staging: thread -> (row = XROWS i + tid / 8, chunk tid % 8).  Rows past m are clamped
  // to the last row: they only feed accumulator rows that are never stored.
  const int srow = tid >> 3, schunk = tid & 7;
  const _Float16* xsrc[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i)
    xsrc[i] = x + (int64_t)min(m0 + srow + XROWS * i, m - 1) * k + 8 * schunk;
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
  const int last_group = (k_end - 1) / group_size, group0 = k_begin / group_size;
  auto load_zs = [&](int grp, ZS& d) {
    grp = min(grp, last_group);
    d.z = zsrc[(int64_t)grp * (n >> 3)];
    d.s = ssrc[(int64_t)grp * (n >> 1)];
  };
  auto load_w = [&](int step, WStep& d) {
    const int k0 = k_begin + phys(step) * TK;
    const uint32_t* p = wsrc + (int64_t)(k0 >> 3) * n;
#pragma unroll
    for (int s = 0; s < 4; ++s) d.w[s] = p[(int64_t)(2 * s) * n];
    if constexpr (!G2) load_zs(k0 / group_size, d.zs);
  };

  // a single 32-row tile would chain all four MFMAs of a step on one accumulator (each waits for
  // the one before): odd slices get a second one there
  constexpr bool DUAL = MT == 1;
  f32x16 acc[MT], acc_odd[DUAL ? 1 : MT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
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
  constexpr int XAHEAD = MT <= 2 ? 4 : 2;
  // G2: zcur holds the constants of step t's group; on even steps znew receives those of the
  // next group (the name the previous group has just released).
  auto k_step = [&](int t, const WStep& wcur, WStep& wnew, XSet& xnew, const XSet& xpark,
                    const ZS& zcur, ZS& znew) {
    const int buf = t & 1;
    load_x(t + XAHEAD, xnew);
    load_w(t + 3, wnew);
    if constexpr (G2) {
      if ((t & 1) == 0) load_zs((k_begin + phys(min(t + 2, nsteps - 2)) * TK) / group_size, znew);
    }
    __builtin_amdgcn_sched_barrier(0);               // keep the loads at the top of the step

    // -(1024 + z) and -(64 + z) as fp16 bit patterns (ulp is 1 resp. 1/16 there, z <= 16)
    const ZS& zs = G2 ? zcur : wcur.zs;
    const uint32_t z = ((zs.z >> zshift) & 15u) + ZOFF;
    const f16x2 zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);
    const f16x2 zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);
    const f16x2 sc2 = as_h2(__builtin_amdgcn_perm(zs.s, zs.s, ssel));

    const unsigned char* xb = xs + buf * X2_TILE_BYTES;
    auto read_a = [&](int s, u32x4 (&af)[MT]) {
#pragma unroll
      for (int a = 0; a < MT; ++a) {
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
    auto mma = [&](const u32x4 (&af)[MT], const u32x4& bf) {
#pragma unroll
      for (int a = 0; a < MT; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[a]),
                                                        __builtin_bit_cast(f16x8, bf), acc[a], 0,
                                                        0, 0);
    };
    auto mma_odd = [&](const u32x4 (&af)[MT], const u32x4& bf) {
      if constexpr (DUAL)
        acc_odd[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]),
                                                            __builtin_bit_cast(f16x8, bf),
                                                            acc_odd[0], 0, 0, 0);
      else
        mma(af, bf);
    };
    // fragments of slice s+1 are requested before the MFMAs of slice s are issued
#define VLA_SB __builtin_amdgcn_sched_barrier(0)
    u32x4 fa[MT], fb[MT];
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
  };

  WStep w0, w1, w2, w3;
  XSet xa, xb_;
  ZS za, zb;                       // G2: groups group0 + 2i in za, group0 + 2i + 1 in zb
  load_x(0, xa);
  load_x(1, xb_);
  int t = 0;
  if constexpr (XAHEAD == 4) {
    XSet xc, xd;                   // x(j) lives in set j % 4
    load_x(2, xc);
    load_x(3, xd);
    if constexpr (G2) load_zs((k_begin + phys(0) * TK) / group_size, za);
    load_w(0, w0);
    load_w(1, w1);
    load_w(2, w2);
    store_x(0, xa);
    __syncthreads();
    for (; t + 4 <= nsteps; t += 4) {
      k_step(t, w0, w3, xa, xb_, za, zb);
      k_step(t + 1, w1, w0, xb_, xc, za, zb);
      k_step(t + 2, w2, w1, xc, xd, zb, za);
      k_step(t + 3, w3, w2, xd, xa, zb, za);
    }
    if (t < nsteps) k_step(t, w0, w3, xa, xb_, za, zb);
    if (t + 1 < nsteps) k_step(t + 1, w1, w0, xb_, xc, za, zb);
    if (t + 2 < nsteps) k_step(t + 2, w2, w1, xc, xd, zb, za);
  } else {
    if constexpr (G2) load_zs((k_begin + phys(0) * TK) / group_size, za);
    load_w(0, w0);
    load_w(1, w1);
    load_w(2, w2);
    store_x(0, xa);
    __syncthreads();
    for (; t + 4 <= nsteps; t += 4) {
      k_step(t, w0, w3, xa, xb_, za, zb);
      k_step(t + 1, w1, w0, xb_, xa, za, zb);
      k_step(t + 2, w2, w1, xa, xb_, zb, za);
      k_step(t + 3, w3, w2, xb_, xa, zb, za);
    }
    if (t < nsteps) k_step(t, w0, w3, xa, xb_, za, zb);
    if (t + 1 < nsteps) k_step(t + 1, w1, w0, xb_, xa, za, zb);
    if (t + 2 < nsteps) k_step(t + 2, w2, w1, xa, xb_, zb, za);
  }

  if constexpr (DUAL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] += acc_odd[0][i];
  }
  // ---- the K parts of the workgroup: parts [stride, 2 stride) hand their sums to parts
  // [0, stride) through LDS (the x buffers, free after the last step's barrier), stride = KW/2 .. 1
  if constexpr (KW > 1) {
    float* red = reinterpret_cast<float*>(xs_all);
    constexpr int SLOT = NW * MT * 16 * 64;        // floats one part hands over
#pragma unroll
    for (int stride = KW / 2; stride >= 1; stride >>= 1) {
      if (wk >= stride && wk < 2 * stride) {
        float* dst = red + (wk - stride) * SLOT + wn * (MT * 16 * 64) + lane;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[(a * 16 + i) * 64] = acc[a][i];
      }
      __syncthreads();
      if (wk < stride) {
        const float* src = red + wk * SLOT + wn * (MT * 16 * 64) + lane;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[a][i] += src[(a * 16 + i) * 64];
      }
      if (stride > 1) __syncthreads();             // the slots are rewritten by the next round
    }
    if (wk != 0) return;
  }
  // ---- epilogue: lane (h, c) holds column ncol, rows (i & 3) + 8 (i >> 2) + 4 h of each 32-row tile
  if (!nok) return;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row >= m) continue;
      if constexpr (SPLIT)
        reinterpret_cast<float*>(outp)[((int64_t)blockIdx.z * slab_rows + row_base + row) * n + ncol] = acc[a][i];
      else
        reinterpret_cast<_Float16*>(outp)[(int64_t)(row_base + row) * n + ncol] = (_Float16)acc[a][i];
    }
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
struct Q4Shape { int nw, splits, mt, kw; };   // kw: K parts inside a workgroup (small batches)

static Q4Shape pick_shape(int m, int n, int k, int group) {
  Q4Shape s;
  s.kw = 1;
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
    if (wgs >= 192) {
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
    s.nw = (n < 3072 || k <= 2048) ? 4 : 8;
    const int wgs = ((n + 32 * s.nw - 1) / (32 * s.nw)) * mtiles;
    want = 256 / wgs;
    if (want > 8) want = 8;
    if (want > k / (8 * TK)) want = k / (8 * TK);
    if (const char* e = getenv("VLA_Q4_MAXSPLIT")) {   // kernel studies
      const int cap = atoi(e);
      if (cap >= 1 && want > cap) want = cap;
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
  }
  const int unit = group > TK ? group : TK;  // a split must hold whole groups and whole K steps
  while (want > 1 && (k % (want * unit)) != 0) --want;
  s.splits = want;
  // K parts inside the workgroup: small row tiles of 128-column workgroups only, whole groups
  // (pairs of K steps where the per-group constants are fetched per pair) in every part
  if (s.mt == 4 || s.nw != 4) kw = 1;
  while (kw > 1 && ((k / want) % (kw * (unit > 2 * TK ? unit : 2 * TK))) != 0) kw >>= 1;
  s.kw = kw;
  return s;
}

extern "C" int64_t vla_q4_gemm_workspace_bytes(int m, int n, int k, int group) {
  if (m <= 0 || n <= 0 || k <= 0 || group <= 0) return 0;
  const int s = pick_shape(m, n, k, group).splits;
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
  Q4Shape shape = pick_shape(m, n, k, group);
  int splits = shape.splits;
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * m * n * 4 ||
                     ((uintptr_t)workspace % 16) || ((int64_t)m * n) % 4))
    splits = 1;
  const int tn = m32 ? 32 * shape.nw : TN, tm = m32 ? 32 * shape.mt : TM;
  dim3 grid((n + tn - 1) / tn, (m + tm - 1) / tm, splits);
  const int kps = k / splits;
  void* dst = splits > 1 ? workspace : out;
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
extern "C" int64_t vla_q4_gemm_grouped_workspace_bytes(int total_rows, int n, int k, int group,
                                                       int num_sets) {
  if (total_rows <= 0 || n <= 0 || k <= 0 || group <= 0 || num_sets <= 0) return 0;
  const int wgs = ((n + 127) / 128) * num_sets;
  int want = (512 + wgs - 1) / wgs;
  if (want > 8) want = 8;
  if (want > k / (4 * TK)) want = k / (4 * TK);
  if (want < 1) want = 1;
  const int unit = group > TK ? group : TK;
  while (want > 1 && (k % (want * unit)) != 0) --want;
  return want > 1 ? (int64_t)want * total_rows * n * 4 : 0;
}

extern "C" int vla_q4_gemm_grouped(const void* x, const int32_t* qweight, const int32_t* qzeros,
                                   const void* scales, void* out, const int* row_offsets,
                                   int num_sets, int total_rows, int max_rows_per_set, int n, int k,
                                   int group, int zero_offset, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  (void)hipGetLastError();
  if (!x || !qweight || !qzeros || !scales || !out || !row_offsets || num_sets <= 0 ||
      total_rows < 0 || max_rows_per_set < 0)
    return VLA_ERR_ARG;
  if (n % 8 || k % TK || group <= 0 || group % TK || k % group ||
      (zero_offset != 0 && zero_offset != 1))
    return VLA_ERR_QUANT_SHAPE;
  if (((uintptr_t)x % 16) || ((uintptr_t)out % 8)) return VLA_ERR_ARG;
  if (total_rows == 0 || max_rows_per_set == 0) return VLA_OK;
  const int64_t need = vla_q4_gemm_grouped_workspace_bytes(total_rows, n, k, group, num_sets);
  int splits = need ? (int)(need / ((int64_t)total_rows * n * 4)) : 1;
  if (splits > 1 && (!workspace || workspace_bytes < need || ((uintptr_t)workspace % 16) ||
                     ((int64_t)total_rows * n) % 4))
    splits = 1;
  Q4Groups gp{};
  gp.row_offsets = row_offsets;
  // 64-row tiles for decode-sized sets, 128-row tiles once the average set fills them
  // (measured on Mixtral shapes: 256 rows per set 1.25 ms with 64-row tiles, 1.31 with 128-row
  // ones; 1024 rows per set 4.26 against 3.99)
  const int mt = total_rows / num_sets >= 384 ? 4 : 2;
  gp.tiles_per_set = (max_rows_per_set + 32 * mt - 1) / (32 * mt);
  gp.total_rows = total_rows;
  gp.w_stride = (int64_t)(k / 8) * n;
  gp.z_stride = (int64_t)(k / group) * (n / 8);
  gp.s_stride = (int64_t)(k / group) * n;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((n + 127) / 128, num_sets * gp.tiles_per_set, splits);
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
  if (splits > 1) {
    // rows of sets that got fewer rows than a tile are never written: the sum must not see junk
    VLA_HIP_CHECK(hipMemsetAsync(workspace, 0, (size_t)need, s));
  }
#define VLA_Q4G_ARGS                                                                            \
  (const _Float16*)x, (const uint32_t*)qweight, (const uint32_t*)qzeros, (const _Float16*)scales, \
      dst, 0, n, k, group, kps, gp
#define VLA_Q4G_G(Z, SP, G2)                                                                    \
  do {                                                                                          \
    if (mt == 4)                                                                                \
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
