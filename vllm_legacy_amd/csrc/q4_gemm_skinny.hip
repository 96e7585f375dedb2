// int4 dequant-GEMM for gfx950, decode batches of at most 16 rows: out[M, N] = x[M, K] . W[K, N].
//
// At these sizes the product is one pass over the weights (HBM-bound: 0.5 byte per weight, two
// flops per row), and what a launch has to get right is bytes in flight and instructions per
// byte -- not matrix-pipe occupancy.  Weight format and arithmetic are those of q4_gemm_fast.hip
// (k-packed pair-order words qw[K/8][N], zeros qz[K/G][N/8], scales [K/G][N]; every weight is
// rebuilt as fp16 (q - z) * s with the reference's rounding, awq/gemm_kernels.cu:156-179,
// gptq/q_gemm.cu:939-947; fp32 accumulation).
//
// Shape of the computation:
//  * mfma_f32_16x16x32_f16 with the WEIGHTS as the A operand (16 columns x 32 k: lane (c, g)
//    holds the k-word g of one column -- exactly one packed word) and x as the B operand (32 k x
//    16 rows: lane (c, g) holds x[c][8g .. 8g+7], one 16-byte load).
//  * a lane fetches FOUR adjacent columns with one 16-byte load (the wave: 4 k-rows x 256
//    contiguous bytes) and issues four MFMAs on them against the same x fragment, so a K step of
//    32 is one weight load, one x load, 52 dequantisation instructions and 4 MFMAs per wave.
//  * x comes straight from global memory (it is a few KiB and L2-resident; rows past M repeat row
//    M - 1 and only feed accumulator columns that are never stored): no LDS tile, no barrier in
//    the loop, every wave runs on its own loads alone.
//  * zeros and scales are fetched once per chunk of CS steps that lies inside one group.
//  * a workgroup is NWV waves on the SAME 64 columns, each walking its own part of the
//    workgroup's K range; their sums meet through LDS once, in wave order (the first four waves
//    each finish and store a quarter of the tile).  Layers with few
//    columns are also cut into K slices across workgroups (fp32 slabs, as the larger kernels).
//
// Measured and not taken: batches of 17-32 rows on the same kernel with a second MFMA per
// dequantised word on rows 16-31 (two x fragments per step, chunks of two steps to stay under 170
// registers): layer (qkv + o + gate_up + down of Llama-3-8B) at batch 32 65.2 us against 59.0 on the
// LDS-staged 32-row kernel, at batch 24 60.9 against 58.1 -- the second x stream from L2 costs
// more than the barrier-free loop saves.  Chunks of two steps (108 registers, four waves per SIMD)
// change nothing at batch 16 either (tools/exp/sweep_skinny_cs.sh).
#include <cstdio>
#include <cstdlib>

#include "vla_common.h"

#ifndef VLA_SK_ABL
#define VLA_SK_ABL 0   // diagnostic builds (wrong results, timing only): 1 no x loads, 2 no dequantisation, 4 no weight loads, 8 default-policy weight loads, 16 strip-major weight addresses
#endif

namespace vla {
namespace {

__device__ __forceinline__ f16x2 sk_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ uint32_t sk_u32(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }

template <int CS>
struct SkChunk {
  u32x4 w[CS];    // four adjacent columns' words of k-row (step, g)
  u32x4 xf[CS];   // x[c][32 step + 8 g ..]
  uint32_t z;     // the zero word holding the lane's four columns
  u32x2 s;        // their four scales
};

// GROUPED: several weight sets in one launch (the experts of an MoE block at a decode batch where
// experts hold a few rows each): blockIdx.y is the set, gp.row_offsets[set] .. [set + 1] its
// rows of x / out, gp.x_rows (optional) the row of x each product row reads; a set without rows
// leaves at once.
struct SkGroups {
  const int* row_offsets;
  const int* x_rows;
  int total_rows;                            // rows of out: the slab pitch of a split product
  int64_t w_stride, z_stride, s_stride;      // elements between consecutive sets
};

// SILU: the product is a [gate | up] pair of halves (a SwiGLU MLP's first projection, n / 2
// columns each) and the launch writes silu(gate) * up, [M, n / 2], instead of the product: a
// workgroup takes 32 gate columns (lanes c < 8) and the 32 up columns that pair with them (lanes
// c >= 8), so after the MFMAs lane (c, g < 2) holds 16 gate values and lane (c, g + 2) the matching
// up values; one cross-half exchange later the activation runs in the epilogue with the
// arithmetic of the stand-alone kernel (both halves rounded to fp16 first, silu rounded, product
// rounded: csrc/activation_kernels.cu:14-33).  Unsliced launches only.
template <int ZOFF, bool SPLIT, int NWV, int CS, int NB, bool GROUPED = false, bool SILU = false>
__global__ __launch_bounds__(64 * NWV) void q4_gemm_skinny_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split, int chunks_per_group,
    SkGroups gp = SkGroups{}) {
  static_assert(NB == 2 || NB == 3, "ring depth");
  __shared__ float red[NWV * 16 * 64];         // [wave][i][j][lane]: every wave's partial sums

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  int row_base = 0, slab_rows = m;
  if constexpr (GROUPED) {
    const int set = blockIdx.y;
    // blockIdx.z = 16-row tile of the set: a set with more than 16 rows streams its weights once
    // per tile (the later ones mostly out of L2 / the Infinity Cache), a tile without rows leaves
    row_base = gp.row_offsets[set] + 16 * blockIdx.z;
    m = min(gp.row_offsets[set + 1] - row_base, 16);
    if (m <= 0) return;
    slab_rows = gp.total_rows;
    qw += (int64_t)set * gp.w_stride;
    qz += (int64_t)set * gp.z_stride;
    scales += (int64_t)set * gp.s_stride;
  }
  static_assert(!SILU || (!SPLIT && !GROUPED), "the fused activation needs the finished product");
  const int n0 = blockIdx.x * 64;
  const int half_n = n >> 1;                   // SILU: n % 16 == 0 (host)
  const int g0 = blockIdx.x * 32;              // SILU: first gate column of this workgroup
  int col, col_end;
  if constexpr (SILU) {
    col = c < 8 ? g0 + 4 * c : half_n + g0 + 4 * (c - 8);
    col_end = c < 8 ? half_n : n;
  } else {
    col = n0 + 4 * c;
    col_end = n;
  }
  const int colc = col < col_end ? col : 0;    // clamped: loads stay in bounds, result unused

  // this wave's chunks of the workgroup's K slice
  constexpr int CK = 32 * CS;
  const int nch_wg = k_per_split / CK;
  const int ch_begin = (wave * nch_wg) / NWV, ch_end = ((wave + 1) * nch_wg) / NWV;
  const int nch = ch_end - ch_begin;
  const int k_begin = (GROUPED ? 0 : blockIdx.z * k_per_split) + ch_begin * CK;

  // uniform bases walk along K in scalar registers; the lane parts never change
#if VLA_SK_ABL & 16   // strip-major addressing: a wave's words of a step are 1 KiB contiguous
  const char* wrow = reinterpret_cast<const char*>(
      qw + ((int64_t)blockIdx.x * (k >> 3) + (k_begin >> 3)) * 64);
  const uint32_t woff = (uint32_t)lane * 16u;
  const int64_t wstep = 1024;
#else
  const char* wrow = reinterpret_cast<const char*>(qw + (int64_t)(k_begin >> 3) * n);
  const uint32_t woff = (uint32_t)(g * n + colc) * 4u;
  const int64_t wstep = (int64_t)n * 16;        // bytes between steps (four k-rows)
#endif
  const char* xrow = reinterpret_cast<const char*>(x + k_begin);
  int xr = row_base + min(c, m - 1);
  if constexpr (GROUPED) {
    if (gp.x_rows) xr = gp.x_rows[xr];
  }
  const uint32_t xoff = (uint32_t)(xr * k + 8 * g) * 2u;
  const int grp0 = k_begin / group_size;
  int grp_ctr = (k_begin - grp0 * group_size) / CK;
  const char* zrow = reinterpret_cast<const char*>(qz + (int64_t)grp0 * (n >> 3));
  const char* srow = reinterpret_cast<const char*>(scales + (int64_t)grp0 * n);
  const uint32_t zoff = (uint32_t)(colc >> 3) * 4u, soff = (uint32_t)colc * 2u;
  const int64_t zstep = (int64_t)(n >> 3) * 4, sstep = (int64_t)n * 2;
  const uint32_t zshift = 16u * (c & 1);

  int issued = 0;
  auto load = [&](SkChunk<CS>& d) {
    if (issued >= nch) return;                  // wave-uniform
    ++issued;
#pragma unroll
    for (int s = 0; s < CS; ++s) {
#if VLA_SK_ABL & 4
      d.w[s] = u32x4{woff, xoff, woff + 1, xoff + 1};
#elif VLA_SK_ABL & 8
      d.w[s] = *reinterpret_cast<const u32x4*>(wrow + woff);
#else
      d.w[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + woff));
#endif
#if VLA_SK_ABL & 1
      d.xf[s] = u32x4{xoff, woff, xoff, woff};
#else
      d.xf[s] = *reinterpret_cast<const u32x4*>(xrow + xoff);
#endif
      wrow += wstep;
      xrow += 64;
    }
    d.z = *reinterpret_cast<const uint32_t*>(zrow + zoff);
    d.s = *reinterpret_cast<const u32x2*>(srow + soff);
    if (++grp_ctr == chunks_per_group) {
      grp_ctr = 0;
      zrow += zstep;
      srow += sstep;
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  // (w & mask) | 0x6400.. as one v_and_or_b32: the exponent pattern sits in a vector register
  // the compiler cannot fold back into a literal
  uint32_t c1024 = 0x64006400u;
  asm volatile("" : "+v"(c1024));
  constexpr uint32_t kLoMask = 0x000f000fu, kHiMask = 0x00f000f0u;

  auto compute = [&](const SkChunk<CS>& d) {
    // -(1024 + z) and -(64 + z) as fp16 bit patterns (ulp is 1 resp. 1/16 there, z <= 16)
    f16x2 zc1[4], zc2[4], sc2[4];
    const uint32_t zw = d.z >> zshift;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t z = ((zw >> (4 * j)) & 15u) + ZOFF;
      zc1[j] = sk_h2(z * 0x00010001u + 0xE400E400u);
      zc2[j] = sk_h2(z * 0x00100010u + 0xD400D400u);
      const uint32_t sw = d.s[j >> 1];
      sc2[j] = sk_h2(__builtin_amdgcn_perm(sw, sw, (j & 1) ? 0x03020302u : 0x01000100u));
    }
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      const f16x8 xb = __builtin_bit_cast(f16x8, d.xf[s]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t w = d.w[s][j], w8 = w >> 8;
        const f16x2 d0 = sk_h2((w & kLoMask) | c1024) + zc1[j];
        const f16x2 d1 = __builtin_elementwise_fma(sk_h2((w & kHiMask) | c1024), sixteenth, zc2[j]);
        const f16x2 d2 = sk_h2((w8 & kLoMask) | c1024) + zc1[j];
        const f16x2 d3 = __builtin_elementwise_fma(sk_h2((w8 & kHiMask) | c1024), sixteenth, zc2[j]);
        u32x4 bf;
#if VLA_SK_ABL & 2
        bf = u32x4{w, w8, w ^ sk_u32(zc1[j]), w8 ^ sk_u32(sc2[j])};
        (void)d0; (void)d1; (void)d2; (void)d3;
#else
        bf[0] = sk_u32(d0 * sc2[j]);
        bf[1] = sk_u32(d1 * sc2[j]);
        bf[2] = sk_u32(d2 * sc2[j]);
        bf[3] = sk_u32(d3 * sc2[j]);
#endif
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf), xb, acc[j], 0,
                                                        0, 0);
      }
    }
  };

  // register ring rotated by name: chunk i lives in buffer i % NB, requested NB - 1 chunks ahead
  SkChunk<CS> b0, b1, b2;
  load(b0);
  if constexpr (NB == 3) load(b1);
  int t = 0;
  if constexpr (NB == 3) {
    for (; t + 3 <= nch; t += 3) {
      load(b2);
      compute(b0);
      load(b0);
      compute(b1);
      load(b1);
      compute(b2);
    }
    if (t < nch) {
      load(b2);
      compute(b0);
    }
    if (t + 1 < nch) compute(b1);
  } else {
    for (; t + 2 <= nch; t += 2) {
      load(b1);
      compute(b0);
      load(b0);
      compute(b1);
    }
    if (t < nch) compute(b0);
  }

  // ---- the K parts of the workgroup meet in LDS: every wave parks its sums, then wave i (of the
  // first four) adds up, in wave order, the quads i of all lanes -- acc[0..3][i], four adjacent
  // columns -- and stores them: the tail of the launch is shared by four waves
  float q[4];                                  // this wave's quad: q[j] = sum of acc[j][qi]
  const int qi = NWV > 1 ? wave : 0;           // which quad this wave finishes (NWV = 1: all four)
  if constexpr (NWV > 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[((wave * 4 + i) * 4 + j) * 64 + lane] = acc[j][i];
    __syncthreads();
    if (wave >= 4) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = red[((0 * 4 + qi) * 4 + j) * 64 + lane];
#pragma unroll
    for (int w = 1; w < NWV; ++w)
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] += red[((w * 4 + qi) * 4 + j) * 64 + lane];
  }
  // lane (c, g) now holds, for x row c, the four adjacent columns 16 g + 4 i + (0..3) of the strip
  // (MFMA j's row r is column 4 r + j)
#pragma unroll
  for (int i0 = 0; i0 < (NWV > 1 ? 1 : 4); ++i0) {
    const int i = NWV > 1 ? qi : i0;
    if constexpr (NWV == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = acc[j][i0];
    }
    if constexpr (SILU) {
      // g < 2: gate columns g0 + 16 g + 4 i + j; g >= 2: the up columns that pair with the gate
      // columns of lane (c, g - 2) = lane - 32
      float up[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) up[j] = __shfl(q[j], (lane + 32) & 63, 64);
      const int oc = g0 + 16 * g + 4 * i;
      if (g < 2 && c < m && oc < half_n) {
        _Float16 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xg = (float)(_Float16)q[j], yu = (float)(_Float16)up[j];
          const float sl = (float)(_Float16)(xg / (1.0f + expf(-xg)));
          r[j] = (_Float16)(sl * yu);
        }
        _Float16* dst = reinterpret_cast<_Float16*>(outp) + (int64_t)c * half_n + oc;
        *reinterpret_cast<u32x2*>(dst) = u32x2{sk_u32(f16x2{r[0], r[1]}), sk_u32(f16x2{r[2], r[3]})};
      }
    } else {
      const int oc = n0 + 16 * g + 4 * i;
      if (c < m && oc < n) {                     // n % 8 == 0: a quad is inside or outside
        if constexpr (SPLIT) {
          float* dst = reinterpret_cast<float*>(outp) +
                       ((int64_t)blockIdx.z * slab_rows + row_base + c) * n + oc;
          *reinterpret_cast<f32x4*>(dst) = f32x4{q[0], q[1], q[2], q[3]};
        } else {
          _Float16* dst = reinterpret_cast<_Float16*>(outp) + (int64_t)(row_base + c) * n + oc;
          const f16x2 lo = {(_Float16)q[0], (_Float16)q[1]};
          const f16x2 hi = {(_Float16)q[2], (_Float16)q[3]};
          *reinterpret_cast<u32x2*>(dst) = u32x2{sk_u32(lo), sk_u32(hi)};
        }
      }
    }
  }
}

}  // namespace

// Launch shape: K slices across workgroups (`splits`, fp32 slabs when > 1) and waves per
// workgroup (`nwv`, K parts summed through LDS).  `cs` = steps per chunk (a chunk lies inside one
// quantisation group).  Returns false when the shape is not covered.
bool q4_skinny_shape(int m, int n, int k, int group, int* nwv, int* splits, int* cs) {
  static const int enabled = [] {
    const char* e = getenv("VLA_Q4_SKINNY");
    return (e && *e) ? atoi(e) : 1;
  }();
  if (!enabled || m < 1 || m > 16) return false;
  if (n % 8 || k % 32 || group <= 0 || group % 32 || k % group) return false;
  if ((int64_t)m * k >= (1ll << 30) || (int64_t)n * 4 >= (1ll << 28)) return false;   // 32-bit lane offsets
  int c = group % 128 == 0 ? 4 : 1;
  if (const char* e = getenv("VLA_Q4_SKINNY_CS")) {   // steps per chunk (kernel studies)
    const int v = atoi(e);
    if ((v == 1 || v == 2 || v == 4) && group % (32 * v) == 0) c = v;
  }
  const int chunks = k / (32 * c);              // per column strip
  const int strips = (n + 63) / 64;
  // Measured at batch 16 on the Llama-3-8B layers (tools/exp/sweep_skinny.sh, us with the
  // reduction launch, "waves per workgroup, K slices"): the launch wants ONE round of workgroups
  // -- a lane holds three chunks of operands in ~156 registers, so a CU takes 12 waves -- and
  // inside that as many waves as there are chunks.
  //  * more strips than CUs (gate_up 4096 x 28672: 448): unsliced, four waves per workgroup, two
  //    or three workgroups per CU: 4,1 18.9 / 8,1 21.7 / 4,2 21.4 / 8,2 27.4;
  //  * otherwise K slices until the grid reaches the CU count (at most four: every slice is an
  //    fp32 slab its consumer reads) and up to 16 waves, a chunk each at least: qkv 4096 x 6144
  //    16,2 8.9 / 8,2 9.0 / 8,4 11.0 (384 workgroups: two rounds); o 4096 x 4096 8,4 7.7 / 16,2
  //    8.5 / 16,1 9.7; down 14336 x 4096 16,4 13.5 / 8,4 13.9 / 16,2 19.2 / 8,8 16.7.
  int w, s;
  if (strips > 256) {
    w = 4;
    s = 1;
  } else {
    s = 256 / strips;
    if (s > 4) s = 4;
    if (s > chunks) s = chunks;
    if (s < 1) s = 1;
    while (s > 1 && chunks % s) --s;
    const int per = chunks / s;
    w = per >= 16 ? 16 : per >= 8 ? 8 : 4;
  }
  if (const char* e = getenv("VLA_Q4_SKINNY_TUNE")) {   // "nwv,splits" (kernel studies)
    int a = 0, b = 0;
    if (sscanf(e, "%d,%d", &a, &b) == 2 && (a == 4 || a == 8 || a == 16) && b >= 1 && b <= 16) {
      w = a;
      s = b;
    }
  }
  while (s > 1 && chunks % s) --s;              // whole chunks in every slice
  *nwv = w;
  *splits = s;
  *cs = c;
  return true;
}

int launch_q4_gemm_skinny(const void* x, const void* qw, const void* qz, const void* scales,
                          void* dst, int m, int n, int k, int group, int zero_offset, int nwv,
                          int splits, int cs, hipStream_t s) {
  static const int nb = [] {
    const char* e = getenv("VLA_Q4_SKINNY_RING");
    return (e && atoi(e) == 2) ? 2 : 3;
  }();
  const dim3 grid((n + 63) / 64, 1, splits);
  const int kps = k / splits;
  const int cpg = group / (32 * cs);
#define VLA_SK_ARGS                                                                              \
  (const _Float16*)x, (const uint32_t*)qw, (const uint32_t*)qz, (const _Float16*)scales, dst, m, \
      n, k, group, kps, cpg
// 1024 threads (W = 16) leave a lane 128 registers: two chunks of four steps, not three
#define VLA_SK_NB(Z, SP, W, C)                                                                   \
  do {                                                                                           \
    if (nb == 2)                                                                                 \
      hipLaunchKernelGGL((q4_gemm_skinny_kernel<Z, SP, W, C, 2>), grid, dim3(64 * W), 0, s,      \
                         VLA_SK_ARGS);                                                           \
    else                                                                                         \
      hipLaunchKernelGGL((q4_gemm_skinny_kernel<Z, SP, W, C, (W == 16 && C == 4) ? 2 : 3>),      \
                         grid, dim3(64 * W), 0, s, VLA_SK_ARGS);                                 \
  } while (0)
#define VLA_SK_W(Z, SP, C)                                                                       \
  do {                                                                                           \
    if (nwv == 16) VLA_SK_NB(Z, SP, 16, C);                                                      \
    else if (nwv == 8) VLA_SK_NB(Z, SP, 8, C);                                                   \
    else VLA_SK_NB(Z, SP, 4, C);                                                                 \
  } while (0)
#define VLA_SK_C(Z, SP)                                                                          \
  do {                                                                                           \
    if (cs == 4) VLA_SK_W(Z, SP, 4);                                                             \
    else if (cs == 2) VLA_SK_W(Z, SP, 2);                                                        \
    else VLA_SK_W(Z, SP, 1);                                                                     \
  } while (0)
  if (zero_offset == 0) {
    if (splits > 1) VLA_SK_C(0, true); else VLA_SK_C(0, false);
  } else {
    if (splits > 1) VLA_SK_C(1, true); else VLA_SK_C(1, false);
  }
#undef VLA_SK_C
#undef VLA_SK_W
#undef VLA_SK_NB
#undef VLA_SK_ARGS
  return 0;
}

// The SwiGLU form (see SILU): whether a [gate | up] product of this shape takes it -- wide enough
// to run unsliced with four waves per workgroup, halves that split into whole 8-column groups.
bool q4_skinny_silu_ok(int m, int n, int k, int group) {
  int nwv = 0, splits = 0, cs = 0;
  if (n % 16 || !q4_skinny_shape(m, n, k, group, &nwv, &splits, &cs)) return false;
  return splits == 1 && nwv == 4 && (cs == 4 || cs == 1);
}

int launch_q4_gemm_skinny_silu(const void* x, const void* qw, const void* qz, const void* scales,
                               void* out, int m, int n, int k, int group, int zero_offset,
                               hipStream_t s) {
  const int cs = group % 128 == 0 ? 4 : 1;
  const dim3 grid((n / 2 + 31) / 32, 1, 1);
  const int cpg = group / (32 * cs);
#define VLA_SKS(Z, C)                                                                            \
  hipLaunchKernelGGL((q4_gemm_skinny_kernel<Z, false, 4, C, 3, false, true>), grid, dim3(256), 0,  \
                     s, (const _Float16*)x, (const uint32_t*)qw, (const uint32_t*)qz,            \
                     (const _Float16*)scales, out, m, n, k, group, k, cpg)
  if (zero_offset == 0) {
    if (cs == 4) VLA_SKS(0, 4); else VLA_SKS(0, 1);
  } else {
    if (cs == 4) VLA_SKS(1, 4); else VLA_SKS(1, 1);
  }
#undef VLA_SKS
  return 0;
}

// Grouped launch (see SkGroups): unsliced, four waves per workgroup -- the shape of the wide dense
// layers, which is what `num_sets` times the column strips of an expert projection amount to.
// row_tiles = ceil(max rows per set / 16).
int launch_q4_gemm_skinny_grouped(const void* x, const int* x_rows, const void* qw, const void* qz,
                                  const void* scales, void* out, const int* row_offsets,
                                  int num_sets, int total_rows, int row_tiles, int n, int k,
                                  int group, int zero_offset, hipStream_t s) {
  const int cs = group % 128 == 0 ? 4 : 1;
  const dim3 grid((n + 63) / 64, num_sets, row_tiles);
  const int cpg = group / (32 * cs);
  SkGroups gp{};
  gp.row_offsets = row_offsets;
  gp.x_rows = x_rows;
  gp.total_rows = total_rows;
  gp.w_stride = (int64_t)(k / 8) * n;
  gp.z_stride = (int64_t)(k / group) * (n / 8);
  gp.s_stride = (int64_t)(k / group) * n;
#define VLA_SKG(Z, C)                                                                            \
  hipLaunchKernelGGL((q4_gemm_skinny_kernel<Z, false, 4, C, 3, true>), grid, dim3(256), 0, s,      \
                     (const _Float16*)x, (const uint32_t*)qw, (const uint32_t*)qz,               \
                     (const _Float16*)scales, out, 0, n, k, group, k, cpg, gp)
  if (zero_offset == 0) {
    if (cs == 4) VLA_SKG(0, 4); else VLA_SKG(0, 1);
  } else {
    if (cs == 4) VLA_SKG(1, 4); else VLA_SKG(1, 1);
  }
#undef VLA_SKG
  return 0;
}

// Whether a grouped product takes the kernel above: few rows per set -- at most VLA_Q4_SKINNY_ROWS
// (default 48) in any set and 16 on average, so that second and third weight passes stay the
// exception -- and enough column strips over all sets to fill the chip without K slices.
bool q4_skinny_grouped_ok(int max_rows_per_set, int total_rows, int num_sets, int n, int k,
                          int group) {
  int nwv = 0, splits = 0, cs = 0;
  static const int max_rows = [] {
    const char* e = getenv("VLA_Q4_SKINNY_ROWS");
    const int v = (e && *e) ? atoi(e) : 48;
    return v < 16 ? 16 : v;
  }();
  if (max_rows_per_set < 1 || max_rows_per_set > max_rows) return false;
  if (max_rows_per_set > 16 && total_rows > 16 * num_sets) return false;
  if (!q4_skinny_shape(1, n, k, group, &nwv, &splits, &cs)) return false;
  if ((int64_t)total_rows * k >= (1ll << 30)) return false;      // 32-bit lane offsets into x
  return (int64_t)((n + 63) / 64) * num_sets > 256;
}

}  // namespace vla
