// int4 dequant-GEMM for gfx950, row blocks of 17..80 rows: out[M, N] = x[M, K] . W[K, N] for decode
// batches between the LDS-free kernel of q4_gemm_skinny.hip (at most 16 rows) and the 128-row
// tiles of q4_gemm_fast.hip, and for the experts of an MoE block that hold a few dozen rows each.
//
// At these sizes one pass over the weights is 0.5 byte and 2 M flops per weight: M = 64 sits on the
// ridge between the HBM and the matrix-pipe rooflines, and what a launch has to get right is the
// vector work per weight -- every weight is rebuilt once (13 instructions per packed word, the
// reference's rounding: awq/gemm_kernels.cu:156-179, gptq/q_gemm.cu:939-947) and then feeds one
// MFMA per 16 rows.  Weight format as in q4_gemm_fast.hip (k-packed pair-order words qw[K/8][N],
// zeros qz[K/G][N/8], scales [K/G][N]), G a multiple of 128.
//
// Shape of the computation (the skinny kernel's, with x through LDS):
//  * mfma_f32_16x16x32_f16, the WEIGHTS as the A operand: lane (c, g) holds k-word g of a step of 32
//    k for four adjacent columns (one nontemporal 16-byte load; the wave: 4 k-rows x 256 contiguous
//    bytes), rebuilds the four words and issues each against the RB row blocks of x: 4 RB MFMAs
//    per weight load.
//  * a workgroup is four waves on four adjacent 64-column strips walking the SAME K range: they
//    share the x tile of a 128-deep chunk (16 RB rows x 256 bytes), which goes global -> registers
//    -> LDS two chunks ahead (a chunk in registers, then one of two buffers): one barrier per
//    chunk.  16-byte piece p of row r sits in slot p ^ (r & 15), so the B-fragment reads
//    (ds_read_b128, rows on lanes c) are conflict-free.
//  * weight words travel in a three-chunk register ring rotated by name; zeros and scales are
//    fetched once per chunk.
//  * the MFMA output puts 16 adjacent columns of one x row on a lane: 32-byte (fp16) or 64-byte
//    (fp32 slab) stores, no LDS in the epilogue.
// Every request of the K loop is unconditional and issued at the head of an iteration (a
// scheduling barrier keeps it there), so the compiler's s_waitcnt counts are exact: parking x
// waits for the x pieces alone and leaves both younger chunks of weight words in flight.
//
// Where the time goes (Mixtral-8x7B w1|w3, 8 experts x 4096 -> 28672, 256 tokens = 64 rows per
// expert, 172-210 us; diagnostic builds -DVLA_MID_ABL / -DVLA_MID_STAMP, counter passes
// tools/exp/pmc_grouped.sh): the vector issue port of a SIMD.  A 16x16x32 MFMA holds the port for 8
// of its 16 cycles, so a chunk costs a wave 920 cycles of vector instructions (13 per packed word)
// plus 8 x 64..80 cycles of MFMA issue, ~1 430-1 560 cycles, beside ~150 of scalar, LDS and
// memory instructions; in-kernel stamps show the two waves of a SIMD each taking ~3 200 cycles per
// chunk (arithmetic 2 140, issuing the ten requests 590 -- the CU's eight waves queue at its
// 64 B/clk texture path --, parking x 260, barrier 120), i.e. 1 600 cycles of SIMD time per wave
// and chunk: the port is full, at the ~1.6 GHz the part holds under this load.  Counters agree
// (matrix pipe 39 % busy, vector ALU 43 %, both through the one port).  What is left is fewer
// instructions per weight -- 32x32x16 MFMAs would halve the MFMA share of the port at the price of
// 32-row granularity (~9 % on this shape, not built); the 13 instructions per word are what the
// reference's rounding costs.  Measured without effect: weight words three or four chunks ahead,
// x through three LDS buffers, eight waves per workgroup (half the x traffic), an MFMA / vector
// interleave by sched_group_barrier, x rows at a non-power-of-two pitch, 32-row tiles in 150
// registers (three waves per SIMD) for launches whose sets hold ~16 rows (64 tokens: 0.189 ->
// 0.184 ms per block).
//
// GROUPED: several weight sets in one launch (blockIdx.y = set, blockIdx.z = 80-row tile of the
// set, row ranges and the optional row gather read on the device); a workgroup runs the body built
// for the number of 16-row blocks its tile holds.
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "vla_common.h"

#ifndef VLA_MID_SCHED
#define VLA_MID_SCHED 0   // 0: leave the issue order of a step to the compiler (kernel studies)
#endif
#ifndef VLA_MID_NB
#define VLA_MID_NB 3      // chunks of weight words in the register ring (kernel studies: 4, 5)
#endif
#ifndef VLA_MID_ABL
#define VLA_MID_ABL 0   // diagnostic builds (wrong results, timing only): 1 no x loads / LDS stores, 2 no dequantisation, 4 no weight loads, 8 one MFMA per word, 16 no barrier, 32 no fragment reads, 64 x rows 272 bytes apart
#endif

namespace vla {
namespace {

__device__ __forceinline__ f16x2 md_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ uint32_t md_u32(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }

constexpr int MID_RBMAX = 5;                    // row blocks of 16 per tile
constexpr int MID_TILE_ROWS = 16 * MID_RBMAX;
constexpr int MID_XBUF = MID_TILE_ROWS * 256;   // bytes of one x buffer (128 k per row)

struct MdChunk {
  u32x4 w[4];     // four adjacent columns' words of k-row (step, g), steps 0..3 of the chunk
  uint32_t z;     // the zero word holding the lane's four columns
  u32x2 s;        // their four scales
};

struct MdGroups {
  const int* row_offsets;
  const int* x_rows;
  int total_rows;                            // rows of out: the slab pitch of a split product
  int splits, k_per_split;                   // K slices: blockIdx.z = row tile * splits + slice
  int64_t w_stride, z_stride, s_stride;      // elements between consecutive sets
};

// One tile: rows [row_base, row_base + m) of x / out (m <= 16 RB), the workgroup's 256 columns, K
// range [k_begin, k_begin + 128 nch).
template <int ZOFF, bool SPLIT, int RB, int NWV>
__device__ __forceinline__ void q4_mid_body(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int row_base, int slab_rows, const int* __restrict__ x_rows, int n, int k,
    int group_size, int k_begin, int nch, int chunks_per_group, int slab, unsigned char* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (64 * NWV) + wave * 64;
  const int col = n0 + 4 * c;
  const int colc = col < n ? col : 0;          // clamped: loads stay in bounds, result unused

  // ---- weights, zeros, scales: uniform bases walk along K, the lane parts never change
  const char* wrow = reinterpret_cast<const char*>(qw + (int64_t)(k_begin >> 3) * n);
  const uint32_t woff = (uint32_t)(g * n + colc) * 4u;
  const int64_t wstep = (int64_t)n * 16;        // bytes between steps (four k-rows)
  const int grp0 = k_begin / group_size;
  int grp_ctr = (k_begin - grp0 * group_size) / 128;
  const char* zrow = reinterpret_cast<const char*>(qz + (int64_t)grp0 * (n >> 3));
  const char* srow = reinterpret_cast<const char*>(scales + (int64_t)grp0 * n);
  const uint32_t zoff = (uint32_t)(colc >> 3) * 4u, soff = (uint32_t)colc * 2u;
  const int64_t zstep = (int64_t)(n >> 3) * 4, sstep = (int64_t)n * 2;
  const uint32_t zshift = 16u * (c & 1);

  // ---- x staging: the tile is 256 RB pieces of 16 bytes, piece p = (row p >> 4, slot p & 15);
  // thread t moves pieces t, t + NT, ... (NT threads)
  constexpr int NT = 64 * NWV;
  constexpr int XP = (256 * RB + NT - 1) / NT;
  const char* xrow = reinterpret_cast<const char*>(x + k_begin);
  uint32_t xoff[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int p = i * NT + tid;
    const int rl = p >> 4, q = p & 15;
    int r = row_base + min(rl, m - 1);
    if (x_rows) r = x_rows[r];
#if VLA_MID_ABL & 64   // rows 272 bytes apart instead of 2 K: the same requests without the power-of-two row pitch
    xoff[i] = (uint32_t)(r * 136 + 8 * (q ^ (rl & 15))) * 2u;
#else
    xoff[i] = (uint32_t)(r * k + 8 * (q ^ (rl & 15))) * 2u;     // rows * K < 2^30 (host check)
#endif
  }
  const uint32_t xst = (uint32_t)tid * 16u;                   // + 16 NT i: where the piece lands
  // B fragments: row 16 rb + c, piece (4 s + g) ^ c
  uint32_t xrd[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) xrd[s] = (uint32_t)(c * 256 + (((4 * s + g) ^ c) << 4));

  // Every request of the loop is unconditional: past the end of the K range the pointers stop
  // advancing and the last chunk is fetched again (and never used).  A branch around a load would
  // cost the compiler its count of outstanding requests, and every wait behind it would become a
  // wait for all of them -- the weight words included, which are meant to stay in flight.
  int w_issued = 0;
  auto loadw = [&](MdChunk& d) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#if VLA_MID_ABL & 4
      d.w[s] = u32x4{woff + s, woff, woff + 1, woff + 2};
#else
      d.w[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + s * wstep + woff));
#endif
    }
    d.z = *reinterpret_cast<const uint32_t*>(zrow + zoff);
    d.s = *reinterpret_cast<const u32x2*>(srow + soff);
    ++w_issued;
    const bool adv = w_issued < nch;            // wave-uniform
    wrow += adv ? 4 * wstep : 0;
    const bool grp = adv && (grp_ctr + 1 == chunks_per_group);
    grp_ctr = adv ? (grp ? 0 : grp_ctr + 1) : grp_ctr;
    zrow += grp ? zstep : 0;
    srow += grp ? sstep : 0;
  };
  u32x4 xs[XP];
  int x_issued = 0;
  auto loadx = [&]() {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
#if VLA_MID_ABL & 1
      xs[i] = u32x4{xoff[i], xoff[i], xoff[i], xoff[i]};
#else
      xs[i] = *reinterpret_cast<const u32x4*>(xrow + xoff[i]);
#endif
    }
    ++x_issued;
    xrow += x_issued < nch ? 256 : 0;
  };
  auto storex = [&](uint32_t buf) {
#if !(VLA_MID_ABL & 1)
#pragma unroll
    for (int i = 0; i < XP; ++i)
      if ((256 * RB) % NT == 0 || i + 1 < XP || i * NT + tid < 256 * RB)
        *reinterpret_cast<u32x4*>(lds + buf + xst + (uint32_t)(16 * NT) * i) = xs[i];
#else
    (void)buf;
#endif
  };

  f32x4 acc[RB][4];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  uint32_t c1024 = 0x64006400u;
  asm volatile("" : "+v"(c1024));
  constexpr uint32_t kLoMask = 0x000f000fu, kHiMask = 0x00f000f0u;

  auto compute = [&](const MdChunk& d, uint32_t buf) {
    f16x2 zc1[4], zc2[4], sc2[4];
    const uint32_t zw = d.z >> zshift;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t z = ((zw >> (4 * j)) & 15u) + ZOFF;
      zc1[j] = md_h2(z * 0x00010001u + 0xE400E400u);
      zc2[j] = md_h2(z * 0x00100010u + 0xD400D400u);
      const uint32_t sw = d.s[j >> 1];
      sc2[j] = md_h2(__builtin_amdgcn_perm(sw, sw, (j & 1) ? 0x03020302u : 0x01000100u));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f16x8 xb[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
#if VLA_MID_ABL & 32
        xb[r] = __builtin_bit_cast(f16x8, u32x4{xrd[s] + buf, xrd[0], xrd[1], (uint32_t)r});
#else
        xb[r] = *reinterpret_cast<const f16x8*>(lds + buf + 4096u * r + xrd[s]);
#endif
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t w = d.w[s][j], w8 = w >> 8;
        const f16x2 d0 = md_h2((w & kLoMask) | c1024) + zc1[j];
        const f16x2 d1 = __builtin_elementwise_fma(md_h2((w & kHiMask) | c1024), sixteenth, zc2[j]);
        const f16x2 d2 = md_h2((w8 & kLoMask) | c1024) + zc1[j];
        const f16x2 d3 = __builtin_elementwise_fma(md_h2((w8 & kHiMask) | c1024), sixteenth, zc2[j]);
        u32x4 bf;
#if VLA_MID_ABL & 2
        bf = u32x4{w, w8, w ^ md_u32(zc1[j]), w8 ^ md_u32(sc2[j])};
        (void)d0; (void)d1; (void)d2; (void)d3;
#else
        bf[0] = md_u32(d0 * sc2[j]);
        bf[1] = md_u32(d1 * sc2[j]);
        bf[2] = md_u32(d2 * sc2[j]);
        bf[3] = md_u32(d3 * sc2[j]);
#endif
        const f16x8 a = __builtin_bit_cast(f16x8, bf);
#pragma unroll
        for (int r = 0; r < ((VLA_MID_ABL & 8) ? 1 : RB); ++r)
          acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[r], acc[r][j], 0, 0, 0);
      }
#if VLA_MID_SCHED
      // issue order of the step: the fragment reads, the first word's dequantisation, then every
      // MFMA followed by a few of the vector instructions that rebuild the NEXT word, so that a
      // wave's own vector work runs in the shadow of its MFMAs (16 cycles each) instead of in
      // front of them
      constexpr int VPER = (39 + 4 * RB - 1) / (4 * RB) + (RB >= 4 ? 1 : 0);
      __builtin_amdgcn_sched_group_barrier(0x100, RB, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 13, 0);
#pragma unroll
      for (int q = 0; q < 4 * RB; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);
      }
#endif
    }
  };

  // ---- chunk t: weights in register ring buffer t % NB, requested NB - 1 chunks ahead; x in LDS
  // buffer t % 2.  The pieces of x(t + 1) are requested at the start of iteration t - 1 (before
  // that iteration's weight request), ride in registers through its arithmetic, are parked in LDS
  // at the start of iteration t and become visible with the barrier that ends it.  Loads return in
  // order: the only weight request older than x(t + 1) is w(t), which iteration t starts with
  // anyway, and w(t + 1), w(t + 2) stay in flight.  (Parking x at the END of the iteration that
  // requested it left the request one chunk's arithmetic to come back, and every wave of the
  // workgroup waited for the slowest at the barrier.)  Buffer (t + 1) % 2 was last read in
  // iteration t - 1, which every wave left through its barrier.
  constexpr uint32_t XB = RB * 4096;
  constexpr int NB = VLA_MID_NB;                // weight ring: NB - 1 chunks ahead
  MdChunk ring[NB];
  loadx();
  loadw(ring[0]);
  storex(0);
  loadx();
#pragma unroll
  for (int u = 1; u < NB - 1; ++u) loadw(ring[u]);
  __syncthreads();
  int t = 0;
  uint32_t rd = 0;
#ifdef VLA_MID_STAMP   // diagnostic build: shader cycles per phase of the K loop, summed per wave
  uint64_t st_park = 0, st_issue = 0, st_math = 0, st_bar = 0;
  const uint64_t st_loop0 = __builtin_amdgcn_s_memtime();
#define VLA_MID_T(v) const uint64_t v = __builtin_amdgcn_s_memtime()
#else
#define VLA_MID_T(v)
#endif
  auto iter = [&](MdChunk& cur, MdChunk& nxt) {
    VLA_MID_T(ta);
    storex(rd ^ XB);
#ifdef VLA_MID_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    VLA_MID_T(tb);
    loadx();
    loadw(nxt);
    __builtin_amdgcn_sched_barrier(0);          // the requests stay ahead of the chunk's arithmetic
    VLA_MID_T(tc);
    compute(cur, rd);
#ifdef VLA_MID_STAMP
    __builtin_amdgcn_sched_barrier(0);
#endif
    VLA_MID_T(td);
#if !(VLA_MID_ABL & 16)
    __syncthreads();
#endif
    VLA_MID_T(te);
#ifdef VLA_MID_STAMP
    st_park += tb - ta; st_issue += tc - tb; st_math += td - tc; st_bar += te - td;
#endif
    rd ^= XB;
    ++t;
  };
  while (t + NB <= nch) {
#pragma unroll
    for (int u = 0; u < NB; ++u) iter(ring[u], ring[(u + NB - 1) % NB]);
  }
#pragma unroll
  for (int u = 0; u < NB - 1; ++u)
    if (t < nch) iter(ring[u], ring[(u + NB - 1) % NB]);

#ifdef VLA_MID_STAMP
  {
    const uint64_t st_loop1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (wave == 0 || wave == NWV - 1) &&
        ((blockIdx.x + 5 * blockIdx.y + 11 * blockIdx.z) % 97) == 0)
      printf("stampmid wg %d,%d,%d wave %d rb %d chunks %d start %llu loop %llu park %llu issue %llu math %llu barrier %llu\n",
             blockIdx.x, blockIdx.y, blockIdx.z, wave, RB, nch,
             (unsigned long long)(st_loop0 % 1000000000ull), (unsigned long long)(st_loop1 - st_loop0),
             (unsigned long long)st_park, (unsigned long long)st_issue, (unsigned long long)st_math,
             (unsigned long long)st_bar);
  }
#endif
  // ---- lane (c, g) holds, for x row 16 r + c, the 16 adjacent columns n0 + 16 g + 4 i + j =
  // acc[r][j][i]
  const int oc = n0 + 16 * g;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int row = 16 * r + c;
    if (row >= m) continue;
    if constexpr (SPLIT) {
      float* dst = reinterpret_cast<float*>(outp) + ((int64_t)slab * slab_rows + row_base + row) * n + oc;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (oc + 4 * i < n)                     // n % 8 == 0: a quad is inside or outside
          *reinterpret_cast<f32x4*>(dst + 4 * i) =
              f32x4{acc[r][0][i], acc[r][1][i], acc[r][2][i], acc[r][3][i]};
    } else {
      _Float16* dst = reinterpret_cast<_Float16*>(outp) + (int64_t)(row_base + row) * n + oc;
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        if (oc + 8 * hlf >= n) continue;        // n % 8 == 0
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 2 * hlf + (e >> 1), j = 2 * (e & 1);
          v[e] = md_u32(f16x2{(_Float16)acc[r][j][i], (_Float16)acc[r][j + 1][i]});
        }
        *reinterpret_cast<u32x4*>(dst + 8 * hlf) = v;
      }
    }
  }
}

template <int ZOFF, bool SPLIT, int RB>
__global__ __launch_bounds__(256, 2) void q4_gemm_mid_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split, int chunks_per_group) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * RB * 4096];
  static_assert(RB <= MID_RBMAX, "tile height");
  q4_mid_body<ZOFF, SPLIT, RB, 4>(x, qw, qz, scales, outp, m, 0, m, nullptr, n, k, group_size,
                               blockIdx.z * k_per_split, k_per_split / 128, chunks_per_group,
                               blockIdx.z, lds);
}

template <int ZOFF, bool SPLIT, int NWV>
__global__ __launch_bounds__(64 * NWV, 2) void q4_gemm_mid_grouped_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int n, int k, int group_size, int chunks_per_group, MdGroups gp) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * MID_XBUF];
  const int set = blockIdx.y;
  const int tile = SPLIT ? blockIdx.z / gp.splits : blockIdx.z;
  const int slice = SPLIT ? blockIdx.z - tile * gp.splits : 0;
  const int row_base = gp.row_offsets[set] + MID_TILE_ROWS * tile;
  const int m = min(gp.row_offsets[set + 1] - row_base, MID_TILE_ROWS);
  if (m <= 0) return;
  qw += (int64_t)set * gp.w_stride;
  qz += (int64_t)set * gp.z_stride;
  scales += (int64_t)set * gp.s_stride;
  const int rb = (m + 15) >> 4;
#define VLA_MID_BODY(RB_)                                                                        \
  q4_mid_body<ZOFF, SPLIT, RB_, NWV>(x, qw, qz, scales, outp, m, row_base, gp.total_rows,        \
                                     gp.x_rows, n, k, group_size, slice * gp.k_per_split,        \
                                     gp.k_per_split / 128, chunks_per_group, slice, lds)
  switch (rb) {
    case 1: VLA_MID_BODY(1); break;
    case 2: VLA_MID_BODY(2); break;
    case 3: VLA_MID_BODY(3); break;
    case 4: VLA_MID_BODY(4); break;
    default: VLA_MID_BODY(5); break;
  }
#undef VLA_MID_BODY
}

}  // namespace

// Where the kernels of this file run: 0 grouped launches only (default: dense products of 17..80
// rows are faster on the tile kernels of q4_gemm_fast.hip, whose launch shapes keep K parts inside
// the workgroup), 1 dense products too (tests, kernel studies), 2 nowhere.  Process-wide, relaxed.
static std::atomic<int> g_mid_mode{[] {
  const char* e = getenv("VLA_Q4_MID");
  const int v = (e && *e) ? atoi(e) : 0;
  return v >= 0 && v <= 2 ? v : 0;
}()};
int q4_mid_set_mode(int mode) {
  if (mode < 0 || mode > 2) return 1;
  g_mid_mode.store(mode, std::memory_order_relaxed);
  return 0;
}

// Whether a dense product of m rows takes this kernel, and in how many K slices (fp32 slabs when
// > 1).  VLA_Q4_MID_SPLITS forces the slice count (kernel studies).
bool q4_mid_shape(int m, int n, int k, int group, int* splits) {
  if (g_mid_mode.load(std::memory_order_relaxed) != 1 || m <= 16 || m > MID_TILE_ROWS) return false;
  if (n % 8 || k % 128 || group <= 0 || group % 128 || k % group) return false;
  if ((int64_t)m * k >= (1ll << 30) || (int64_t)n * 4 >= (1ll << 28)) return false;   // 32-bit lane offsets
  const int wgs = (n + 255) / 256;
  const int chunks = k / 128;
  int s = 1;
  if (wgs < 256) {
    // K slices until the launch has about two workgroups per CU, at least four chunks each
    s = (512 + wgs - 1) / wgs;
    if (s > 16) s = 16;
    if (s > chunks / 4) s = chunks / 4;
    if (s < 1) s = 1;
  }
  if (const char* e = getenv("VLA_Q4_MID_SPLITS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 32) s = v;
  }
  while (s > 1 && (chunks % s || (k / s) % group)) --s;        // whole chunks and groups per slice
  *splits = s;
  return true;
}

int launch_q4_gemm_mid(const void* x, const void* qw, const void* qz, const void* scales, void* dst,
                       int m, int n, int k, int group, int zero_offset, int splits, hipStream_t s) {
  const dim3 grid((n + 255) / 256, 1, splits);
  const int kps = k / splits;
  const int cpg = group / 128;
  const int rb = (m + 15) / 16;
#define VLA_MID_ARGS                                                                             \
  (const _Float16*)x, (const uint32_t*)qw, (const uint32_t*)qz, (const _Float16*)scales, dst, m, \
      n, k, group, kps, cpg
#define VLA_MID_RB(Z, SP)                                                                        \
  do {                                                                                           \
    switch (rb) {                                                                                \
      case 2: hipLaunchKernelGGL((q4_gemm_mid_kernel<Z, SP, 2>), grid, dim3(256), 0, s, VLA_MID_ARGS); break; \
      case 3: hipLaunchKernelGGL((q4_gemm_mid_kernel<Z, SP, 3>), grid, dim3(256), 0, s, VLA_MID_ARGS); break; \
      case 4: hipLaunchKernelGGL((q4_gemm_mid_kernel<Z, SP, 4>), grid, dim3(256), 0, s, VLA_MID_ARGS); break; \
      default: hipLaunchKernelGGL((q4_gemm_mid_kernel<Z, SP, 5>), grid, dim3(256), 0, s, VLA_MID_ARGS); break; \
    }                                                                                            \
  } while (0)
  if (zero_offset == 0) {
    if (splits > 1) VLA_MID_RB(0, true); else VLA_MID_RB(0, false);
  } else {
    if (splits > 1) VLA_MID_RB(1, true); else VLA_MID_RB(1, false);
  }
#undef VLA_MID_RB
#undef VLA_MID_ARGS
  return 0;
}

// Grouped launch.  K slices (fp32 slabs [slice][total_rows][n], summed by the caller's reduction
// launch) when the sets' column workgroups alone do not fill the chip: an expert's second
// projection is long and narrow (Mixtral-8x7B w2: 14336 -> 4096, 16 workgroups per expert).
int q4_mid_grouped_splits(int total_rows, int num_sets, int n, int k, int group) {
  // Rows per set on average up to which the launch takes this kernel (a set above 80 rows streams
  // its weights once per 80-row tile).  Mixtral-8x7B block from a graph, this kernel against the
  // 128-row tiles of q4_gemm_fast.hip: 64 tokens (16 rows per expert) 0.213 / 0.234 ms, 256 tokens
  // 0.286 / 0.365, 512 tokens 0.501 / 0.595, 768 tokens 0.744 / 0.762, 1024 tokens 0.939 / 0.957,
  // 2048 tokens 1.77 / 1.73.
  static const int max_avg = [] {
    const char* e = getenv("VLA_Q4_MID_ROWS");
    return (e && *e) ? atoi(e) : 200;
  }();
  if (g_mid_mode.load(std::memory_order_relaxed) == 2 || total_rows < 1 || num_sets < 1) return 0;
  if (n % 8 || k % 128 || group <= 0 || group % 128 || k % group) return 0;
  if ((int64_t)total_rows * k >= (1ll << 30) || (int64_t)n * 4 >= (1ll << 28)) return 0;
  if ((int64_t)total_rows > (int64_t)max_avg * num_sets) return 0;
  const int64_t wgs = (int64_t)((n + 255) / 256) * num_sets;
  const int chunks = k / 128;
  int s = 1;
  if (wgs < 256) {
    s = (int)((512 + wgs - 1) / wgs);
    if (s > 8) s = 8;
    if (s > chunks / 8) s = chunks / 8;
    if (s < 1) s = 1;
    while (s > 1 && (chunks % s || (k / s) % group)) --s;
    if (wgs * s < 128) return 0;               // too small a launch for this kernel
  }
  if (const char* e = getenv("VLA_Q4_MID_GSPLITS")) {   // kernel studies
    const int v = atoi(e);
    if (v >= 1 && v <= 16 && chunks % v == 0 && (k / v) % group == 0) s = v;
  }
  return s;
}

int launch_q4_gemm_mid_grouped(const void* x, const int* x_rows, const void* qw, const void* qz,
                               const void* scales, void* dst, const int* row_offsets, int num_sets,
                               int total_rows, int max_rows_per_set, int n, int k, int group,
                               int zero_offset, int splits, hipStream_t s) {
  // VLA_Q4_MID_NWV=4 / 8: waves (64-column strips) per workgroup that share an x tile
  static const int nwv = [] {
    const char* e = getenv("VLA_Q4_MID_NWV");
    const int v = (e && *e) ? atoi(e) : 4;
    return v == 8 ? 8 : 4;
  }();
  const int tiles = (max_rows_per_set + MID_TILE_ROWS - 1) / MID_TILE_ROWS;
  const dim3 grid((n + 64 * nwv - 1) / (64 * nwv), num_sets, tiles * splits);
  MdGroups gp{};
  gp.row_offsets = row_offsets;
  gp.x_rows = x_rows;
  gp.total_rows = total_rows;
  gp.splits = splits;
  gp.k_per_split = k / splits;
  gp.w_stride = (int64_t)(k / 8) * n;
  gp.z_stride = (int64_t)(k / group) * (n / 8);
  gp.s_stride = (int64_t)(k / group) * n;
  const int cpg = group / 128;
#define VLA_MIDG(Z, SP, W)                                                                       \
  hipLaunchKernelGGL((q4_gemm_mid_grouped_kernel<Z, SP, W>), grid, dim3(64 * W), 0, s,           \
                     (const _Float16*)x, (const uint32_t*)qw, (const uint32_t*)qz,               \
                     (const _Float16*)scales, dst, n, k, group, cpg, gp)
#define VLA_MIDG_W(Z, SP)                                                                        \
  do {                                                                                           \
    if (nwv == 8) VLA_MIDG(Z, SP, 8); else VLA_MIDG(Z, SP, 4);                                   \
  } while (0)
  if (zero_offset == 0) {
    if (splits > 1) VLA_MIDG_W(0, true); else VLA_MIDG_W(0, false);
  } else {
    if (splits > 1) VLA_MIDG_W(1, true); else VLA_MIDG_W(1, false);
  }
#undef VLA_MIDG_W
#undef VLA_MIDG
  return 0;
}

}  // namespace vla
