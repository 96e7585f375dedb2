// Causal self-attention over packed prompt tokens (prefill) for gfx950, 16-bit models.
//
// Replaces what the reference computes for prompts without a cached prefix
// (vllm/attention/backends/rocm_flash_attn.py:305-383: Triton / CK flash attention or the naive
// masked path; semantics as ref_masked_attention, tests/kernels/test_attention.py:43-55):
// per sequence s with tokens [cu[s], cu[s+1]) and per head, out_i = softmax_j<=i(scale q_i.k_j
// + slope (j - i)) v_j, with grouped KV heads.
//
// One 256-thread workgroup per (128-query tile, head, sequence); wave w owns queries
// 32w .. 32w+31.  Keys/values are walked in tiles of 64 staged through LDS (shared by the four
// waves).  Both products run on 32x32x16 MFMAs in transposed form, lane = (h, c) = (lane >> 5,
// lane & 31):
//   S^T tile:  A = K   lane(h, c): key c of a 32-key block, dims 16 ks + 8h .. + 7   (LDS, b128)
//              B = q^T lane(h, c): query c, the same dims                            (registers)
//              C       lane(h, c): query c, keys (i & 3) + 8 (i >> 2) + 4h, i = 0..15
//   O^T tile:  A = V^T lane(h, c): dim 32n + c, keys 16 sl + 8h .. + 7
//                      (row-major V image, two ds_read_b64_tr_b16; cached V is stored that way)
//              B = P^T lane(h, c): query c, the same keys (registers)
//              C       lane(h, c): query c, dims 32n + (i & 3) + 8 (i >> 2) + 4h
// so a lane holds one query's logits, running max/sum and output column: the softmax is a
// reduction over registers and one exchange with the partner lane (h ^ 1), rescaling needs no
// exchange, and the probabilities go from the C layout of the first product to the B operand of
// the second with one v_permlane32_swap per 8 keys, never through LDS.
// Two forms share that arithmetic: prefill_attention_kernel (256 threads as above, two workgroups
// per CU; short query runs pack 2 or 4 heads of a KV group into the tile) and
// prefill_attention_il_kernel (512 threads: two such sub-tiles share every staged K/V tile, one
// workgroup per CU, the tile loop one tile ahead on the key side); the launcher picks by shape
// (pf_use_il), vla_set_prefill_form overrides.
// Numerics: fp32 logits and running max/sum, exp values rounded to the model dtype before the
// value product, fp32 accumulation, one division at the end.
#include "vla_common.h"

#include <math.h>
#include <stdlib.h>

#include <atomic>

namespace vla {
namespace {

constexpr int PF_BM = 128, PF_BN = 64, PF_THREADS = 256, PF2_THREADS = 512;
#ifndef VLA_PF_ABL   // timing studies: bit mask of parts left out (wrong results)
#define VLA_PF_ABL 0
#endif

template <class S>
__device__ __forceinline__ f32x16 pf_mfma32(u32x4 a, u32x4 b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 pf_mfma32<F16>(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 pf_mfma32<BF16>(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// two fp32 -> one word of two model-dtype values (round to nearest even), as one packed convert
template <class S>
__device__ __forceinline__ uint32_t pf_pack2(float a, float b);
template <>
__device__ __forceinline__ uint32_t pf_pack2<F16>(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
template <>
__device__ __forceinline__ uint32_t pf_pack2<BF16>(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

struct PrefillParams {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  const int* cu_seqlens;          // [num_seqs + 1] start of each sequence's new tokens
  const float* alibi_slopes;
  int num_heads, num_kv_heads;
  int64_t q_stride, k_stride, v_stride, out_stride;
  float scale;
  int sliding_window;             // 0 = off
  // cached prefix (PREFIX kernels only)
  const void* k_cache;
  const void* v_cache;
  const int* block_tables;
  const int* seq_lens;            // [num_seqs] cached + new tokens
  const int* ctx_lens;            // [num_seqs] cached tokens
  int max_blocks_per_seq;
  int log2_block_size;
  int64_t kv_block_stride, kv_head_stride;
  float kv_scale;
  int xcd_groups;                 // heads x sequences is a multiple of 8 KV groups: XCD-aware order
  int pair_heads;                 // 512-thread form: a workgroup takes two heads of a KV group
  int heads_per_tile;             // 256-thread form: 1, 2 or 4 heads of a KV group share a tile
};

// 16-byte chunk swizzles of the LDS images (RB bytes per row):
//   K image (row = key) and the transposed V image of cached blocks (row = dim), read by
//            ds_read_b128 with rows on lanes: 256-byte rows chunk ^ (row & 15); 128-byte rows
//            chunk ^ ((row >> 1) & 7) -- two rows share a 256-byte bank line, the row parity
//            picks its half, so 16 consecutive rows land on 16 distinct 16-byte slots
//   V image (row = key), read by ds_read_b64_tr_b16 (8 keys x 32 bytes per half wave): 256-byte
//            rows chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)); 128-byte rows
//            chunk ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2))
template <int DP>
__device__ __forceinline__ int k_swz(int row, int ch) {
  return DP == 128 ? (ch ^ (row & 15)) : (ch ^ ((row >> 1) & 7));
}
template <int DP>
__device__ __forceinline__ int v_swz(int row, int ch) {
  return DP == 128 ? (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)))
                   : (ch ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2)));
}

// 16 fp8 bytes -> 16 model-dtype values (two 16-byte chunks).  scale == 1 takes the exact
// hardware conversion; otherwise round(float(fp8) * scale) per element, as the decode kernels'
// reference does (csrc/quantization/fp8/amd/quant_utils.cuh:310-318).
template <class S>
__device__ __forceinline__ void pf_fp8x16(const u32x4 w, float scale, u32x4& lo, u32x4& hi) {
  uint32_t r[8];
  if (scale == 1.0f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x2 t = fp8x4_to_16<S>(w[i]);
      r[2 * i] = t[0];
      r[2 * i + 1] = t[1];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 a = fp8x2_to_f32<false>(w[i]), b = fp8x2_to_f32<true>(w[i]);
      r[2 * i] = pf_pack2<S>(a[0] * scale, a[1] * scale);
      r[2 * i + 1] = pf_pack2<S>(b[0] * scale, b[1] * scale);
    }
  }
  lo = u32x4{r[0], r[1], r[2], r[3]};
  hi = u32x4{r[4], r[5], r[6], r[7]};
}

// keep the first `live` (0..8) 16-bit elements of a chunk, zero the rest
__device__ __forceinline__ u32x4 pf_keep_first(u32x4 w, int live) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t m = (2 * e < live ? 0xffffu : 0u) | (2 * e + 1 < live ? 0xffff0000u : 0u);
    w[e] &= m;
  }
  return w;
}

// D: head size (a multiple of 16, at most 128; 112 needs the registers of a whole SIMD lane
// budget, so it runs one workgroup per CU); the LDS images are laid out for DP = 64 or 128
// dims per row, the query fragments and the K image zero-padded up to a multiple of 32 dims.
// PREFIX: keys [0, ctx) of a sequence come from the paged cache (FP8: as e4m3 bytes).
// WIN: a sliding window is in force (p.sliding_window > 0).
template <class S, int D, bool PREFIX, bool FP8, bool WIN>
__global__ __launch_bounds__(PF_THREADS, D == 112 ? 1 : 2) void prefill_attention_kernel(PrefillParams p) {
  using T = typename S::T;
  constexpr int DP = D <= 64 ? 64 : 128;
  constexpr int RB = DP * 2;                // bytes per K/V row in LDS
  constexpr int CH = D / 8;                 // live 16-byte chunks per row
  constexpr int KS = (D + 15) / 16;         // 16-deep MFMA slices of the S^T product (the last one
                                            // half zeros when D is an odd number of 8-dim chunks)
  constexpr int ND = (D + 31) / 32;         // 32-dim output tiles (the last one may be half used)
  constexpr int LD = (PF_BN * CH + PF_THREADS - 1) / PF_THREADS;  // 16-byte loads per thread, tile
  constexpr bool LD_EXACT = (PF_BN * CH) % PF_THREADS == 0;
  static_assert(D % 8 == 0 && D <= 128 && D >= 16, "head size");
  static_assert(!(FP8 && D % 16), "a byte cache packs 16 dims per 16-byte K piece");
  static_assert(PF_BN * RB >= D * PF_BN * 2, "transposed V image fits");
  __shared__ __attribute__((aligned(16))) unsigned char k_lds[PF_BN * RB];
  __shared__ __attribute__((aligned(16))) unsigned char v_lds[PF_BN * RB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, c = lane & 31;
  // grid = (heads, sequences, query tiles): workgroups are handed out x-fastest, so all heads
  // and sequences of one query tile go together -- the heads of a KV group side by side -- and
  // the tiles with the longest key walks (the last ones) are handed out first
  // Workgroups go to the 8 XCDs round robin in that order, and each XCD has its own L2: the
  // heads of a KV group, which walk the same keys and values at the same time, are therefore
  // placed 8 apart in dispatch order -- on one XCD -- and every (sequence, KV head) stream is
  // read through a single L2, whatever the query tile.
  // Short query runs (a chunk of a prompt behind a long cached prefix, a few tokens to verify):
  // 2 or 4 heads of a KV group share the 128-row tile, 64 or 32 queries each, instead of leaving
  // half or three quarters of its waves without queries (p.heads_per_tile).
  const int hpt = p.heads_per_tile, seg_rows = PF_BM / hpt, wps = 4 / hpt;
  int hw = blockIdx.x, seq = blockIdx.y;
  const int gsz = p.num_heads / p.num_kv_heads, gx = gsz / hpt;
  if (p.xcd_groups) {
    const int r = blockIdx.x + gridDim.x * blockIdx.y;
    const int slot = r >> 3, grp = (slot / gx) * 8 + (r & 7);
    seq = grp / p.num_kv_heads;
    hw = (grp % p.num_kv_heads) * gx + slot % gx;
  }
  const int head = hw * hpt + wave / wps;
  const int wrow = 32 * (wave % wps);        // this wave's first query inside its head's rows
  const int kv_head = head / gsz;
  const int seq_start = p.cu_seqlens[seq];
  const int ctx = PREFIX ? p.ctx_lens[seq] : 0;
  // number of new tokens; positions are absolute: query i sits at ctx + i
  const int n_new = PREFIX ? p.seq_lens[seq] - ctx : p.cu_seqlens[seq + 1] - seq_start;
  const int q0 = (gridDim.z - 1 - blockIdx.z) * seg_rows;
  if (q0 >= n_new) return;
  // a window known to be absent at compile time keeps its tests out of the tile loop (5 % of
  // the kernel's time when left to run time)
  const int W = WIN ? p.sliding_window : 0;

  const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)seq_start * p.q_stride + (int64_t)head * D;
  const T* kb = reinterpret_cast<const T*>(p.k) + (int64_t)seq_start * p.k_stride + (int64_t)kv_head * D;
  const T* vb = reinterpret_cast<const T*>(p.v) + (int64_t)seq_start * p.v_stride + (int64_t)kv_head * D;

  // ---- query fragments: lane (h, c) holds dims 16 s + 8 h .. + 7 of query c; rows past the
  // sequence are clamped (their results are never stored)
  u32x4 qf[KS];
  const int qrow = q0 + wrow + c;
  const int qpos = ctx + qrow;
  {
    const T* qp = qb + (int64_t)min(qrow, n_new - 1) * p.q_stride;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      qf[ks] = (D % 16 == 0 || 2 * ks + h < CH) ? *reinterpret_cast<const u32x4*>(qp + 16 * ks + 8 * h)
                                                : u32x4{0, 0, 0, 0};
  }
  if constexpr (D % 16 != 0) {
    // the chunk past the head in the last slice of every K row: never written by a tile, read
    // against zero query dims -- it must hold zeros, not whatever the LDS held (0 * NaN)
    if (tid < PF_BN)
      *reinterpret_cast<u32x4*>(k_lds + tid * RB + 16 * k_swz<DP>(tid, CH)) = u32x4{0, 0, 0, 0};
  }
  constexpr float kLog2e = 1.4426950408889634f;
  const float scale2 = p.scale * kLog2e;
  const float slope2 = p.alibi_slopes ? p.alibi_slopes[head] * kLog2e : 0.f;
  // positions of this wave's first and last query
  const int wq_min = ctx + q0 + wrow, wq_max = wq_min + 31;

  // O^T accumulators: lane (h, c) holds dims 32 n + (i & 3) + 8 (i >> 2) + 4 h of query c
  f32x16 o[ND];
#pragma unroll
  for (int n = 0; n < ND; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[n][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  u32x4 kreg[LD], vreg[LD];

  // ---- key tiles of 64: first the cached prefix [0, ctx) out of the paged cache, then the new
  // tokens up to the last query of this workgroup (causal).  K blocks are [D/x][BS][x] and V
  // blocks [D][BS], each contiguous, so consecutive threads take consecutive 16-byte pieces of a
  // block; cached K lands row-major (row = key) like new keys, cached V as it is stored
  // (row = dim).
  const int npre = PREFIX ? (ctx + PF_BN - 1) / PF_BN : 0;
  const int ntiles = npre + (min(n_new, q0 + seg_rows) + PF_BN - 1) / PF_BN;
  const int lbs = PREFIX ? p.log2_block_size : 0, BS = 1 << lbs;
  const int* bt = PREFIX ? p.block_tables + (int64_t)seq * p.max_blocks_per_seq : nullptr;
  const int last_blk = ctx > 0 ? (ctx - 1) >> lbs : 0;
  constexpr int EB = FP8 ? 1 : 2;                       // bytes per cached element
  constexpr int KX = 16 / EB;                           // dims per 16-byte piece of cached K
  constexpr int NPIECE = PF_BN * D / KX;                // pieces per cached tile (K or V)
  constexpr int LP = (NPIECE + PF_THREADS - 1) / PF_THREADS;
  static_assert(LP <= LD, "piece registers");
  const unsigned char* kcb = nullptr;
  const unsigned char* vcb = nullptr;
  if (PREFIX) {
    kcb = reinterpret_cast<const unsigned char*>(p.k_cache) + (int64_t)kv_head * p.kv_head_stride * EB;
    vcb = reinterpret_cast<const unsigned char*>(p.v_cache) + (int64_t)kv_head * p.kv_head_stride * EB;
  }
  const int tvl = lbs - (FP8 ? 4 : 3);                  // log2(16-byte pieces per V row of a block)

  // byte offsets of this thread's pieces inside a whole tile of new tokens (the host bounds the
  // strides)
  uint32_t koff[LD], voff[LD];
#pragma unroll
  for (int i = 0; i < LD; ++i) {
    const int id = tid + PF_THREADS * i, row = id / CH, ch = id % CH;
    koff[i] = (uint32_t)row * (uint32_t)p.k_stride * 2u + 16u * ch;
    voff[i] = (uint32_t)row * (uint32_t)p.v_stride * 2u + 16u * ch;
  }

#ifdef VLA_PF_STAMP   // diagnostic build: shader cycles per phase of the tile loop, summed per wave
  uint64_t sp_load = 0, sp_stage = 0, sp_qk = 0, sp_sm = 0, sp_pv = 0;
  uint64_t sp_tb = 0;
  int sp_n = 0;
  const uint64_t sp_t0 = __builtin_amdgcn_s_memtime();
#define VLA_PF_T(v) __builtin_amdgcn_sched_barrier(0); const uint64_t v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define VLA_PF_T(v)
#endif
  for (int it = 0; it < ntiles; ++it) {
    const bool pre = PREFIX && it < npre;               // workgroup-uniform
    VLA_PF_T(ta);
    // first key of the tile (absolute position) and the end of the keys it may hold
    const int kpos0 = pre ? it * PF_BN : ctx + (it - npre) * PF_BN;
    const int limit = pre ? ctx : ctx + n_new;
    // the whole tile lies before the window of this workgroup's first query
    if (W > 0 && ctx + q0 - (kpos0 + PF_BN - 1) >= W) continue;
    // two workgroups share a CU: while this one waits for its tile, the other one computes
    if (pre) {
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int id = tid + PF_THREADS * i;
        if (NPIECE % PF_THREADS == 0 || id < NPIECE) {
          // K piece: [block in tile][dim group][token in block]
          const int off = id & (BS - 1), rest = id >> lbs;
          const int cg = rest % (D / KX), blk = rest / (D / KX);
          const int64_t nb = bt[min((kpos0 >> lbs) + blk, last_blk)];
          kreg[i] = *reinterpret_cast<const u32x4*>(kcb + nb * p.kv_block_stride * EB +
                                                    ((int64_t)(cg * BS + off) * 16));
          // V piece: [block in tile][dim][token group]
          const int tg = id & ((1 << tvl) - 1), rest2 = id >> tvl;
          const int d = rest2 % D, blk2 = rest2 / D;
          const int64_t nb2 = bt[min((kpos0 >> lbs) + blk2, last_blk)];
          vreg[i] = *reinterpret_cast<const u32x4*>(vcb + nb2 * p.kv_block_stride * EB +
                                                    ((int64_t)d * BS * EB + tg * 16));
        }
      }
      __syncthreads();                 // every wave is done reading the previous tile
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int id = tid + PF_THREADS * i;
        if (NPIECE % PF_THREADS == 0 || id < NPIECE) {
          const int off = id & (BS - 1), rest = id >> lbs;
          const int cg = rest % (D / KX), blk = rest / (D / KX);
          const int row = (blk << lbs) + off;
          const int tg = id & ((1 << tvl) - 1), rest2 = id >> tvl;
          const int d = rest2 % D, blk2 = rest2 / D;
          // first key of the V piece and how many of its keys exist
          const int key0 = (blk2 << lbs) + tg * (16 / EB);
          const int live = ctx - (kpos0 + key0);
          unsigned char* vrow = v_lds + d * (PF_BN * 2);
          if (FP8) {
            u32x4 lo, hi;
            pf_fp8x16<S>(kreg[i], p.kv_scale, lo, hi);
            *reinterpret_cast<u32x4*>(k_lds + row * RB + 16 * k_swz<DP>(row, 2 * cg)) = lo;
            *reinterpret_cast<u32x4*>(k_lds + row * RB + 16 * k_swz<DP>(row, 2 * cg + 1)) = hi;
            pf_fp8x16<S>(vreg[i], p.kv_scale, lo, hi);
            if (live < 16) {
              lo = pf_keep_first(lo, max(live, 0));
              hi = pf_keep_first(hi, max(live - 8, 0));
            }
            *reinterpret_cast<u32x4*>(vrow + 16 * k_swz<64>(d, key0 >> 3)) = lo;
            *reinterpret_cast<u32x4*>(vrow + 16 * k_swz<64>(d, (key0 >> 3) + 1)) = hi;
          } else {
            *reinterpret_cast<u32x4*>(k_lds + row * RB + 16 * k_swz<DP>(row, cg)) = kreg[i];
            u32x4 w = vreg[i];
            // a value past the prefix may be anything, NaN included: its weight is zero, so must it be
            if (live < 8) w = pf_keep_first(w, max(live, 0));
            *reinterpret_cast<u32x4*>(vrow + 16 * k_swz<64>(d, key0 >> 3)) = w;
          }
        }
      }
    } else {
      // a whole tile's address is scalar arithmetic plus the per-thread piece offsets; only a
      // sequence's last, partial tile clamps rows
      const int kt = it - npre, rows = n_new - kt * PF_BN;
      const unsigned char* kt_base = reinterpret_cast<const unsigned char*>(kb) + (int64_t)kt * PF_BN * p.k_stride * 2;
      const unsigned char* vt_base = reinterpret_cast<const unsigned char*>(vb) + (int64_t)kt * PF_BN * p.v_stride * 2;
      if (rows >= PF_BN) {
#pragma unroll
        for (int i = 0; i < LD; ++i)
          if (LD_EXACT || tid + PF_THREADS * i < PF_BN * CH) {
            kreg[i] = *reinterpret_cast<const u32x4*>(kt_base + koff[i]);
            vreg[i] = *reinterpret_cast<const u32x4*>(vt_base + voff[i]);
          }
      } else {
#pragma unroll
        for (int i = 0; i < LD; ++i) {
          const int id = tid + PF_THREADS * i, row = id / CH, ch = id % CH;
          if (LD_EXACT || id < PF_BN * CH) {
            const int64_t tok = min(row, rows - 1);
            kreg[i] = *reinterpret_cast<const u32x4*>(kt_base + tok * p.k_stride * 2 + 16 * ch);
            vreg[i] = *reinterpret_cast<const u32x4*>(vt_base + tok * p.v_stride * 2 + 16 * ch);
          }
        }
      }
      __syncthreads();                 // every wave is done reading the previous tile
#ifdef VLA_PF_STAMP
      __builtin_amdgcn_sched_barrier(0);
      sp_tb = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int i = 0; i < LD; ++i) {
        const int id = tid + PF_THREADS * i, row = id / CH, ch = id % CH;
        if (LD_EXACT || id < PF_BN * CH) {
          *reinterpret_cast<u32x4*>(k_lds + row * RB + 16 * k_swz<DP>(row, ch)) = kreg[i];
          *reinterpret_cast<u32x4*>(v_lds + row * RB + 16 * v_swz<DP>(row, ch)) = vreg[i];
        }
      }
    }
    __syncthreads();
    VLA_PF_T(tc);
    // wave-uniform: the wave has no query at all (a short last tile), every key follows every
    // query of this wave, or lies before all their windows
    if (q0 + wrow >= n_new) continue;
    if (kpos0 > wq_max) continue;
    if (W > 0 && wq_min - (kpos0 + PF_BN - 1) >= W) continue;

    // ---- S^T = K q^T for the tile's two 32-key blocks: lane (h, c) gets, of query c, keys
    // 32 b + (i & 3) + 8 (i >> 2) + 4 h
    f32x16 st[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) st[b][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int row = 32 * b + c;
        const u32x4 kf = *reinterpret_cast<const u32x4*>(k_lds + row * RB +
                                                         16 * k_swz<DP>(row, 2 * ks + h));
        st[b] = pf_mfma32<S>(kf, qf[ks], st[b]);
      }
    VLA_PF_T(td);

    // ---- online softmax of query c over this lane's 32 keys and its partner lane's, in base 2:
    // logits are scaled by scale * log2(e) so that every exponential is a bare v_exp_f32.  A
    // tile that lies entirely below the diagonal of this wave's queries, inside the sequence and
    // inside every window needs no mask (wave-uniform test).
    const bool interior = (kpos0 + PF_BN - 1 <= wq_min) && (kpos0 + PF_BN <= limit) &&
                          (W <= 0 || wq_max - kpos0 < W);
    // Unmasked tiles without ALiBi (most of a long prompt) take the maximum of the raw products
    // and fold the scale into the exponent's multiply-add; scale > 0 keeps the maximum where it
    // is.  The two cases are separate code all the way to the packed probabilities: merged into
    // selects, the masking arithmetic would run on every tile.
    const bool plain = interior && slope2 == 0.f && scale2 > 0.f;
    float m_new, alpha, psum = 0.f;
    // probabilities rounded to the model dtype, packed in key order: pw[b][j] = keys
    // 32 b + 8 j + 4 h + 0 .. 3 of query c (two words)
    u32x2 pw[2][4];
    // exponent of element x: x * es - m_ref
    auto finish = [&](float mx, const float es) {
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      m_new = fmaxf(m_run, mx);
      // a query with no live key so far keeps m = -inf: use 0 as the reference point then
      const float m_ref = m_new == -INFINITY ? 0.f : m_new;
      alpha = __builtin_amdgcn_exp2f(m_run - m_ref);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float e[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[b][4 * j + i], es, -m_ref));
            psum += e[i];
          }
          pw[b][j][0] = pf_pack2<S>(e[0], e[1]);
          pw[b][j][1] = pf_pack2<S>(e[2], e[3]);
        }
    };
    if (plain) {
      float mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[b][i]);
      finish(mx * scale2, scale2);
    } else {
      float mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = kpos0 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h;
          float x = st[b][i] * scale2;
          if (slope2 != 0.f) x += slope2 * (float)(key - qpos);
          if (!interior) {
            bool live = key <= qpos && key < limit;
            if (W > 0) live = live && (qpos - key < W);
            x = live ? x : -INFINITY;
          }
          st[b][i] = x;
          mx = fmaxf(mx, x);
        }
      finish(mx, 1.f);
    }
    l_run = l_run * alpha + psum;          // this lane's keys only; the halves are added at the end
    if (__any(m_new > m_run)) {
#pragma unroll
      for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[n][i] *= alpha;
    }
    m_run = m_new;
    VLA_PF_T(te);

    // ---- O^T += V^T P^T, 16 keys at a time.  The B operand wants, in lane (h, c), keys
    // 16 s + 8 h .. + 7 of query c: the lane's own four keys of that group of eight and its
    // partner's four, which one v_permlane32_swap of the two groups of a slice delivers.
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int b = sl >> 1, j0 = 2 * (sl & 1);
      u32x4 pf;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const auto r = __builtin_amdgcn_permlane32_swap(pw[b][j0][w], pw[b][j0 + 1][w], false, false);
        pf[w] = r[0];        // keys 8 h' + 0 .. 3 of the slice (two per word)
        pf[2 + w] = r[1];    // keys 8 h' + 4 .. 7
      }
#pragma unroll
      for (int n = 0; n < ND; ++n) {
        u32x4 vf;
        if (pre) {
          // cached tile: row = dim 32 n + c, 128-byte rows of 64 keys, keys 16 sl + 8 h .. + 7 are
          // one chunk
          const int row = 32 * n + c;
          vf = *reinterpret_cast<const u32x4*>(v_lds + row * (PF_BN * 2) +
                                               16 * k_swz<64>(row, 2 * sl + h));
        } else {
          // row-major tile: each group of 16 lanes reads 4 keys x 16 dims and receives them
          // transposed, one dim per lane
          const int i16 = c & 15;
          const int r0 = 16 * sl + 8 * h + (i16 >> 2);
          const int ch = 4 * n + 2 * (c >> 4) + ((i16 & 3) >> 1), half = (i16 & 1) * 8;
          typedef short s16x4 __attribute__((ext_vector_type(4)));
          typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_ptr)(v_lds + r0 * RB + 16 * v_swz<DP>(r0, ch) + half));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_ptr)(v_lds + (r0 + 4) * RB + 16 * v_swz<DP>(r0 + 4, ch) + half));
          const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
          vf[0] = lo2[0]; vf[1] = lo2[1]; vf[2] = hi2[0]; vf[3] = hi2[1];
        }
        o[n] = pf_mfma32<S>(vf, pf, o[n]);
      }
    }
#ifdef VLA_PF_STAMP
    {
      VLA_PF_T(tf);
      sp_load += sp_tb - ta; sp_stage += tc - sp_tb; sp_qk += td - tc; sp_sm += te - td; sp_pv += tf - te;
      ++sp_n;
    }
#endif
  }
#ifdef VLA_PF_STAMP
  {
    const uint64_t sp_t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (blockIdx.x % 16) == 0 && blockIdx.y == 0 && (blockIdx.z % 8) == 0)
      printf("stamppf wg %d,%d,%d wave %d tiles %d of %d total %llu issue+bar %lld wait+stage+bar %llu qk %llu softmax %llu pv %llu\n",
             blockIdx.x, blockIdx.y, blockIdx.z, wave, sp_n, ntiles, (unsigned long long)(sp_t1 - sp_t0),
             (long long)sp_load, (unsigned long long)sp_stage, (unsigned long long)sp_qk,
             (unsigned long long)sp_sm, (unsigned long long)sp_pv);
  }
#endif

  // ---- normalise and store: lane (h, c) holds dims 32 n + (i & 3) + 8 (i >> 2) + 4 h of query c
  l_run += __shfl_xor(l_run, 32, 64);
  if (qrow < n_new) {
    T* op = reinterpret_cast<T*>(p.out) + (int64_t)(seq_start + qrow) * p.out_stride + (int64_t)head * D;
    const float inv = 1.f / l_run;
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d0 = 32 * n + 8 * j + 4 * h;
        if (D % 32 == 0 || d0 < D) {
          u32x2 w;
          w[0] = pf_pack2<S>(o[n][4 * j] * inv, o[n][4 * j + 1] * inv);
          w[1] = pf_pack2<S>(o[n][4 * j + 2] * inv, o[n][4 * j + 3] * inv);
          *reinterpret_cast<u32x2*>(op + d0) = w;
        }
      }
  }
}

// The 512-thread form.  Eight waves share each staged K/V tile: waves 0-3 and 4-7 are two
// 128-query sub-tiles -- two heads of one KV group at the same query positions when the group
// size is even (p.pair_heads; both halves then have the same shape), otherwise two consecutive
// query tiles of one head -- so a tile's bytes are read from L2 once per 256 queries.  The tile
// loop is one tile ahead on the key side: while tile t is in its softmax the wave's matrix pipe
// already runs S^T of tile t + 1 (one basic block, so the softmax's vector instructions issue
// between the MFMAs of the next product), then O^T of tile t.  LDS holds K of tile t + 1 beside
// V of tile t; K(t + 2) and V(t + 1) are requested before the second product and stored after it.
// The query fragments live in LDS (the registers they would take hold the logits that are one
// tile ahead).  LDS images have padded rows instead of swizzled chunks, so that every read address
// of the tile loop is one per-lane base plus a compile-time offset.
template <class S, int D, bool PREFIX, bool FP8, bool WIN>
__global__ __launch_bounds__(PF2_THREADS) void prefill_attention_il_kernel(PrefillParams p) {
  using T = typename S::T;
  constexpr int DP = D <= 64 ? 64 : 128;
  constexpr int RB = DP * 2;                // bytes of the dims of a K/V row in LDS
  constexpr int CH = D / 8;                 // live 16-byte chunks per row
  constexpr int KS = (D + 15) / 16;         // 16-deep MFMA slices of the S^T product (the last one
                                            // half zeros when D is an odd number of 8-dim chunks)
  constexpr int ND = (D + 31) / 32;         // 32-dim output tiles (the last one may be half used)
  constexpr int LD = (PF_BN * CH + PF2_THREADS - 1) / PF2_THREADS;  // 16-byte loads per thread, tile
  constexpr bool LD_EXACT = (PF_BN * CH) % PF2_THREADS == 0;
  static_assert(D % 8 == 0 && D <= 128 && D >= 16, "head size");
  static_assert(!(FP8 && D % 16), "a byte cache packs 16 dims per 16-byte K piece");
  // K and q rows (read as b128, rows on lanes) are 16 bytes longer than a power of two, V rows
  // (read transposed, four rows x 64 bytes per half wave) 64 bytes longer; the cached-V image
  // (row = dim, 64 keys) like K.  All conflict-free.
  constexpr int KP = RB + 16, VP = RB + 64, CP = PF_BN * 2 + 16;
  constexpr int V_BYTES = PF_BN * VP > D * CP ? PF_BN * VP : D * CP;
  __shared__ __attribute__((aligned(16))) unsigned char k_lds[PF_BN * KP];
  __shared__ __attribute__((aligned(16))) unsigned char v_lds[V_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char q_lds[2 * PF_BM * KP];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = wave >> 2, w4 = wave & 3;
  const int h = lane >> 5, c = lane & 31;
  // grid = (heads or head pairs, sequences, query tiles): workgroups are handed out x-fastest,
  // so all heads and sequences of one query tile go together, and the tiles with the longest
  // key walks (the last ones) are handed out first.  Workgroups go to the 8 XCDs round robin in
  // that order, and each XCD has its own L2: the workgroups of a KV group, which walk the same
  // keys and values at the same time, are therefore placed 8 apart in dispatch order -- on one
  // XCD -- and every (sequence, KV head) stream is read through a single L2, whatever the
  // query tile.
  const int hx = p.pair_heads ? 2 : 1;                 // heads per workgroup
  const int gsz = p.num_heads / p.num_kv_heads, gx = gsz / hx;
  int hw = blockIdx.x, seq = blockIdx.y;
  if (p.xcd_groups) {
    const int r = blockIdx.x + gridDim.x * blockIdx.y;
    const int slot = r >> 3, grp = (slot / gx) * 8 + (r & 7);
    seq = grp / p.num_kv_heads;
    hw = (grp % p.num_kv_heads) * gx + slot % gx;
  }
  const int head = p.pair_heads ? 2 * hw + sub : hw;
  const int kv_head = head / gsz;
  const int seq_start = p.cu_seqlens[seq];
  const int ctx = PREFIX ? p.ctx_lens[seq] : 0;
  // number of new tokens; positions are absolute: query i sits at ctx + i
  const int n_new = PREFIX ? p.seq_lens[seq] - ctx : p.cu_seqlens[seq + 1] - seq_start;
  const int zq = gridDim.z - 1 - blockIdx.z;
  // first query of the workgroup, of this wave's sub-tile, and the end of the workgroup's queries
  const int q0_wg = p.pair_heads ? zq * PF_BM : 2 * zq * PF_BM;
  const int q0 = p.pair_heads ? q0_wg : q0_wg + sub * PF_BM;
  const int q_end = p.pair_heads ? q0_wg + PF_BM : q0_wg + 2 * PF_BM;
  if (q0_wg >= n_new) return;
  // a window known to be absent at compile time keeps its tests out of the tile loop
  const int W = WIN ? p.sliding_window : 0;

  const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)seq_start * p.q_stride + (int64_t)head * D;
  const unsigned char* kb = reinterpret_cast<const unsigned char*>(
      reinterpret_cast<const T*>(p.k) + (int64_t)seq_start * p.k_stride + (int64_t)kv_head * D);
  const unsigned char* vb = reinterpret_cast<const unsigned char*>(
      reinterpret_cast<const T*>(p.v) + (int64_t)seq_start * p.v_stride + (int64_t)kv_head * D);

  // ---- query fragments: lane (h, c) reads dims 16 s + 8 h .. + 7 of query c as the B operand;
  // rows past the sequence are clamped (their results are never stored).  Each wave writes and
  // reads its own 32 rows of the q image.
  const int qrow = q0 + 32 * w4 + c;
  const int qpos = ctx + qrow;
  const unsigned char* q_row = q_lds + (32 * wave + c) * KP + 16 * h;
  {
    const T* qp = qb + (int64_t)max(min(qrow, n_new - 1), 0) * p.q_stride;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 w = (D % 16 == 0 || 2 * ks + h < CH)
                          ? *reinterpret_cast<const u32x4*>(qp + 16 * ks + 8 * h) : u32x4{0, 0, 0, 0};
      *reinterpret_cast<u32x4*>(q_lds + (32 * wave + c) * KP + 16 * (2 * ks + h)) = w;
    }
  }
  if constexpr (D % 16 != 0) {
    // the chunk past the head in the last slice of every K row: never written by a tile, read
    // against zero query dims -- it must hold zeros, not whatever the LDS held (0 * NaN)
    if (tid < PF_BN)
      *reinterpret_cast<u32x4*>(k_lds + tid * KP + 16 * CH) = u32x4{0, 0, 0, 0};
  }
  constexpr float kLog2e = 1.4426950408889634f;
  const float scale2 = p.scale * kLog2e;
  const float slope2 = p.alibi_slopes ? p.alibi_slopes[head] * kLog2e : 0.f;
  // positions of this wave's first and last query
  const int wq_min = ctx + q0 + 32 * w4, wq_max = wq_min + 31;

  // O^T accumulators: lane (h, c) holds dims 32 n + (i & 3) + 8 (i >> 2) + 4 h of query c
  f32x16 o[ND];
#pragma unroll
  for (int n = 0; n < ND; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[n][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  u32x4 kreg[LD], vreg[LD];

  // ---- key tiles of 64: first the cached prefix [0, ctx) out of the paged cache, then the new
  // tokens up to the last query of this workgroup (causal).  K blocks are [D/x][BS][x] and V
  // blocks [D][BS], each contiguous, so consecutive threads take consecutive 16-byte pieces of a
  // block; cached K lands row-major (row = key) like new keys, cached V as it is stored
  // (row = dim).
  const int npre = PREFIX ? (ctx + PF_BN - 1) / PF_BN : 0;
  const int ntiles = npre + (min(n_new, q_end) + PF_BN - 1) / PF_BN;
  const int lbs = PREFIX ? p.log2_block_size : 0, BS = 1 << lbs;
  const int* bt = PREFIX ? p.block_tables + (int64_t)seq * p.max_blocks_per_seq : nullptr;
  const int last_blk = ctx > 0 ? (ctx - 1) >> lbs : 0;
  constexpr int EB = FP8 ? 1 : 2;                       // bytes per cached element
  constexpr int KX = 16 / EB;                           // dims per 16-byte piece of cached K
  constexpr int NPIECE = PF_BN * D / KX;                // pieces per cached tile (K or V)
  constexpr int LP = (NPIECE + PF2_THREADS - 1) / PF2_THREADS;
  static_assert(LP <= LD, "piece registers");
  const unsigned char* kcb = nullptr;
  const unsigned char* vcb = nullptr;
  if (PREFIX) {
    kcb = reinterpret_cast<const unsigned char*>(p.k_cache) + (int64_t)kv_head * p.kv_head_stride * EB;
    vcb = reinterpret_cast<const unsigned char*>(p.v_cache) + (int64_t)kv_head * p.kv_head_stride * EB;
  }
  const int tvl = lbs - (FP8 ? 4 : 3);                  // log2(16-byte pieces per V row of a block)
  // byte offsets of this thread's pieces inside a whole tile of new tokens: the tile's address is
  // scalar arithmetic, a piece's address that plus one of these (the host bounds the strides)
  uint32_t koff[LD], voff[LD];
#pragma unroll
  for (int i = 0; i < LD; ++i) {
    const int id = tid + PF2_THREADS * i, row = id / CH, ch = id % CH;
    koff[i] = (uint32_t)row * (uint32_t)p.k_stride * 2u + 16u * ch;
    voff[i] = (uint32_t)row * (uint32_t)p.v_stride * 2u + 16u * ch;
  }

  auto kpos_of = [&](int it) { return (PREFIX && it < npre) ? it * PF_BN : ctx + (it - npre) * PF_BN; };
  const bool wave_has_q = q0 + 32 * w4 < n_new;
  // wave-uniform: some query of this wave sees some key of the tile
  auto wave_active = [&](int it) {
    const int kp = kpos_of(it);
    return wave_has_q && kp <= wq_max && !(W > 0 && wq_min - (kp + PF_BN - 1) >= W);
  };
  // new-token tile kt: `dst` <- this thread's pieces of rows [64 kt, 64 kt + 64) of a [token][stride]
  // array, rows past the sequence clamped to its last one
  auto load_new = [&](u32x4 (&dst)[LD], const unsigned char* b0, int64_t stride, const uint32_t (&off)[LD],
                      int kt) {
    const unsigned char* base = b0 + (int64_t)kt * PF_BN * stride * 2;
    const int rows = n_new - kt * PF_BN;
    if (rows >= PF_BN) {
#pragma unroll
      for (int i = 0; i < LD; ++i)
        if (LD_EXACT || tid + PF2_THREADS * i < PF_BN * CH)
          dst[i] = *reinterpret_cast<const u32x4*>(base + off[i]);
    } else {
#pragma unroll
      for (int i = 0; i < LD; ++i) {
        const int id = tid + PF2_THREADS * i, row = id / CH, ch = id % CH;
        if (LD_EXACT || id < PF_BN * CH)
          dst[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)min(row, rows - 1) * stride * 2 + 16 * ch);
      }
    }
  };
  auto load_k = [&](int it, u32x4 (&kreg)[LD]) {
    if (PREFIX && it < npre) {
      const int kp = it * PF_BN;
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int id = tid + PF2_THREADS * i;
        if (NPIECE % PF2_THREADS == 0 || id < NPIECE) {
          // K piece: [block in tile][dim group][token in block]
          const int off = id & (BS - 1), rest = id >> lbs;
          const int cg = rest % (D / KX), blk = rest / (D / KX);
          const int64_t nb = bt[min((kp >> lbs) + blk, last_blk)];
          kreg[i] = *reinterpret_cast<const u32x4*>(kcb + nb * p.kv_block_stride * EB +
                                                    ((int64_t)(cg * BS + off) * 16));
        }
      }
    } else {
      load_new(kreg, kb, p.k_stride, koff, it - npre);
    }
  };
  auto load_v = [&](int it) {
    if (PREFIX && it < npre) {
      const int kp = it * PF_BN;
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int id = tid + PF2_THREADS * i;
        if (NPIECE % PF2_THREADS == 0 || id < NPIECE) {
          // V piece: [block in tile][dim][token group]
          const int tg = id & ((1 << tvl) - 1), rest2 = id >> tvl;
          const int d = rest2 % D, blk2 = rest2 / D;
          const int64_t nb2 = bt[min((kp >> lbs) + blk2, last_blk)];
          vreg[i] = *reinterpret_cast<const u32x4*>(vcb + nb2 * p.kv_block_stride * EB +
                                                    ((int64_t)d * BS * EB + tg * 16));
        }
      }
    } else {
      load_new(vreg, vb, p.v_stride, voff, it - npre);
    }
  };
  auto store_k = [&](int it, const u32x4 (&kreg)[LD]) {
    if (PREFIX && it < npre) {
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int id = tid + PF2_THREADS * i;
        if (NPIECE % PF2_THREADS == 0 || id < NPIECE) {
          const int off = id & (BS - 1), rest = id >> lbs;
          const int cg = rest % (D / KX), blk = rest / (D / KX);
          const int row = (blk << lbs) + off;
          if (FP8) {
            u32x4 lo, hi;
            pf_fp8x16<S>(kreg[i], p.kv_scale, lo, hi);
            *reinterpret_cast<u32x4*>(k_lds + row * KP + 16 * (2 * cg)) = lo;
            *reinterpret_cast<u32x4*>(k_lds + row * KP + 16 * (2 * cg + 1)) = hi;
          } else {
            *reinterpret_cast<u32x4*>(k_lds + row * KP + 16 * cg) = kreg[i];
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < LD; ++i) {
        const int id = tid + PF2_THREADS * i, row = id / CH, ch = id % CH;
        if (LD_EXACT || id < PF_BN * CH)
          *reinterpret_cast<u32x4*>(k_lds + row * KP + 16 * ch) = kreg[i];
      }
    }
  };
  auto store_v = [&](int it) {
    if (PREFIX && it < npre) {
      const int kp = it * PF_BN;
#pragma unroll
      for (int i = 0; i < LP; ++i) {
        const int id = tid + PF2_THREADS * i;
        if (NPIECE % PF2_THREADS == 0 || id < NPIECE) {
          const int tg = id & ((1 << tvl) - 1), rest2 = id >> tvl;
          const int d = rest2 % D, blk2 = rest2 / D;
          // first key of the V piece and how many of its keys exist
          const int key0 = (blk2 << lbs) + tg * (16 / EB);
          const int live = ctx - (kp + key0);
          unsigned char* vrow = v_lds + d * CP;
          if (FP8) {
            u32x4 lo, hi;
            pf_fp8x16<S>(vreg[i], p.kv_scale, lo, hi);
            if (live < 16) {
              lo = pf_keep_first(lo, max(live, 0));
              hi = pf_keep_first(hi, max(live - 8, 0));
            }
            *reinterpret_cast<u32x4*>(vrow + 16 * (key0 >> 3)) = lo;
            *reinterpret_cast<u32x4*>(vrow + 16 * ((key0 >> 3) + 1)) = hi;
          } else {
            u32x4 w = vreg[i];
            // a value past the prefix may be anything, NaN included: its weight is zero, so must it be
            if (live < 8) w = pf_keep_first(w, max(live, 0));
            *reinterpret_cast<u32x4*>(vrow + 16 * (key0 >> 3)) = w;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < LD; ++i) {
        const int id = tid + PF2_THREADS * i, row = id / CH, ch = id % CH;
        if (LD_EXACT || id < PF_BN * CH)
          *reinterpret_cast<u32x4*>(v_lds + row * VP + 16 * ch) = vreg[i];
      }
    }
  };
  // S^T = K q^T for the two 32-key blocks of the tile in LDS: lane (h, c) gets, of query c, keys
  // 32 b + (i & 3) + 8 (i >> 2) + 4 h
  auto qk = [&](f32x16 (&st)[2]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 qf = *reinterpret_cast<const u32x4*>(q_row + 32 * ks);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(k_lds + (32 * b + c) * KP + 16 * (2 * ks + h));
        if (!(VLA_PF_ABL & 16) || ks == 0) st[b] = pf_mfma32<S>(kf, qf, ks == 0 ? zero : st[b]);
      }
    }
  };

  // one tile: `st` holds its S^T (when this wave takes part in it), `nx` receives the next one's
  auto tile = [&](int it, f32x16 (&st)[2], f32x16 (&nx)[2]) {
    const bool pre = PREFIX && it < npre;               // workgroup-uniform
    // first key of the tile (absolute position) and the end of the keys it may hold
    const int kpos0 = kpos_of(it);
    const int limit = pre ? ctx : ctx + n_new;
    const bool act = wave_active(it), act_n = it + 1 < ntiles && wave_active(it + 1);
    // the next tiles' pieces are in flight over all of this tile's arithmetic: one workgroup
    // owns the CU, nobody else would cover their latency (past the end the last tile is loaded
    // and stored again: the same bytes in the same place)
    if (!(VLA_PF_ABL & 4)) {
      load_k(min(it + 2, ntiles - 1), kreg);
      load_v(min(it + 1, ntiles - 1));
    }

    // ---- online softmax of query c over this lane's 32 keys and its partner lane's, in base 2:
    // logits are scaled by scale * log2(e) so that every exponential is a bare v_exp_f32.  A
    // tile that lies entirely below the diagonal of this wave's queries, inside the sequence and
    // inside every window needs no mask (wave-uniform test).
    const bool interior = (kpos0 + PF_BN - 1 <= wq_min) && (kpos0 + PF_BN <= limit) &&
                          (W <= 0 || wq_max - kpos0 < W);
    // Unmasked tiles without ALiBi (most of a long prompt) take the maximum of the raw products
    // and fold the scale into the exponent's multiply-add (scale > 0 keeps the maximum where it
    // is).  The others are scaled, biased and masked in place first, so that one and the same
    // softmax block follows: it is the block the next tile's first product is issued in.
    const bool plain = interior && slope2 == 0.f && scale2 > 0.f;
    float es = scale2;                // exponent of element x: x * es - m_ref
    if (act && !plain) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = kpos0 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h;
          float x = st[b][i] * scale2;
          if (slope2 != 0.f) x += slope2 * (float)(key - qpos);
          if (!interior) {
            bool live = key <= qpos && key < limit;
            if (W > 0) live = live && (qpos - key < W);
            x = live ? x : -INFINITY;
          }
          st[b][i] = x;
        }
      es = 1.f;
    }
    float m_new = m_run, alpha = 1.f, psum = 0.f;
    // probabilities rounded to the model dtype, packed in key order: pw[b][j] = keys
    // 32 b + 8 j + 4 h + 0 .. 3 of query c (two words)
    u32x2 pw[2][4];
    if (act || act_n) {
      // (a wave in its last tile computes a product nobody uses, against whatever tile the K
      // image holds; a wave that only starts with the next tile a softmax nobody uses)
      qk(nx);
      if (VLA_PF_ABL & 1) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pw[b][j][0] = pf_pack2<S>(st[b][4 * j], st[b][4 * j + 1]);
            pw[b][j][1] = pf_pack2<S>(st[b][4 * j + 2], st[b][4 * j + 3]);
          }
      } else {
        float mx = -INFINITY;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[b][i]);
        mx *= es;
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        m_new = fmaxf(m_run, mx);
        // a query with no live key so far keeps m = -inf: use 0 as the reference point then
        const float m_ref = m_new == -INFINITY ? 0.f : m_new;
        alpha = __builtin_amdgcn_exp2f(m_run - m_ref);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[b][4 * j + i], es, -m_ref));
              psum += e[i];
            }
            pw[b][j][0] = pf_pack2<S>(e[0], e[1]);
            pw[b][j][1] = pf_pack2<S>(e[2], e[3]);
          }
      }
    }
    if (act) {
      l_run = l_run * alpha + psum;          // this lane's keys only; the halves are added at the end
      if (__any(m_new > m_run)) {
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[n][i] *= alpha;
      }
      m_run = m_new;

      // ---- O^T += V^T P^T, 16 keys at a time.  The B operand wants, in lane (h, c), keys
      // 16 s + 8 h .. + 7 of query c: the lane's own four keys of that group of eight and its
      // partner's four, which one v_permlane32_swap of the two groups of a slice delivers.
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const int b = sl >> 1, j0 = 2 * (sl & 1);
        u32x4 pf;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const auto r = __builtin_amdgcn_permlane32_swap(pw[b][j0][w], pw[b][j0 + 1][w], false, false);
          pf[w] = r[0];        // keys 8 h' + 0 .. 3 of the slice (two per word)
          pf[2 + w] = r[1];    // keys 8 h' + 4 .. 7
        }
#pragma unroll
        for (int n = 0; n < ND; ++n) {
          u32x4 vf;
          if (pre) {
            // cached tile: row = dim 32 n + c, rows of 64 keys, keys 16 sl + 8 h .. + 7 are one chunk
            vf = *reinterpret_cast<const u32x4*>(v_lds + (32 * n + c) * CP + 16 * (2 * sl + h));
          } else {
            // row-major tile: each group of 16 lanes reads 4 keys x 16 dims and receives them
            // transposed, one dim per lane
            const int i16 = c & 15;
            const int r0 = 16 * sl + 8 * h + (i16 >> 2);
            const int ch = 4 * n + 2 * (c >> 4) + ((i16 & 3) >> 1), half = (i16 & 1) * 8;
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_ptr)(v_lds + r0 * VP + 16 * ch + half));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_ptr)(v_lds + (r0 + 4) * VP + 16 * ch + half));
            const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
            vf[0] = lo2[0]; vf[1] = lo2[1]; vf[2] = hi2[0]; vf[3] = hi2[1];
          }
          if (!(VLA_PF_ABL & 2) || (sl == 0 && n == 0)) o[n] = pf_mfma32<S>(vf, pf, o[n]);
        }
      }
    }
    if (!(VLA_PF_ABL & 8))
      __syncthreads();                 // every wave is done with K of tile it + 1 and V of tile it
    if (!(VLA_PF_ABL & 4)) {
      store_k(min(it + 2, ntiles - 1), kreg);
      store_v(min(it + 1, ntiles - 1));
    }
    if (!(VLA_PF_ABL & 8))
      __syncthreads();
  };

  // tiles that lie before the window of this workgroup's first query are a prefix of the walk
  int it0 = 0;
  if (WIN)
    while (it0 + 1 < ntiles && W > 0 && ctx + q0_wg - (kpos_of(it0) + PF_BN - 1) >= W) ++it0;
  f32x16 sa[2], sb[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) sa[b][i] = sb[b][i] = 0.f;
  {
    // the first three tile images are requested together: one load latency, not two
    u32x4 k0[LD];
    load_k(it0, k0);
    load_k(min(it0 + 1, ntiles - 1), kreg);
    load_v(it0);
    store_k(it0, k0);
  }
  __syncthreads();
  if (wave_active(it0)) qk(sa);
  __syncthreads();                     // every wave is done with K of the first tile
  store_k(min(it0 + 1, ntiles - 1), kreg);
  store_v(it0);
  __syncthreads();
  for (int it = it0; it < ntiles; it += 2) {
    tile(it, sa, sb);
    if (it + 1 < ntiles) tile(it + 1, sb, sa);
  }

  // ---- normalise and store: lane (h, c) holds dims 32 n + (i & 3) + 8 (i >> 2) + 4 h of query c
  l_run += __shfl_xor(l_run, 32, 64);
  if (qrow < n_new) {
    T* op = reinterpret_cast<T*>(p.out) + (int64_t)(seq_start + qrow) * p.out_stride + (int64_t)head * D;
    const float inv = 1.f / l_run;
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d0 = 32 * n + 8 * j + 4 * h;
        if (D % 32 == 0 || d0 < D) {
          u32x2 w;
          w[0] = pf_pack2<S>(o[n][4 * j] * inv, o[n][4 * j + 1] * inv);
          w[1] = pf_pack2<S>(o[n][4 * j + 2] * inv, o[n][4 * j + 3] * inv);
          *reinterpret_cast<u32x2*>(op + d0) = w;
        }
      }
  }
}

}  // namespace
}  // namespace vla

using namespace vla;

namespace {

bool pf_head_size_ok(int d) {
  return d == 24 || d == 64 || d == 80 || d == 96 || d == 112 || d == 128;
}

// the XCD-aware workgroup order needs whole KV groups on each of the 8 XCDs in every row of
// (workgroups per sequence) x sequences workgroups
int pf_xcd_groups(int wgs_per_seq, int wgs_per_group, int num_seqs) {
  static const bool off = [] { const char* e = getenv("VLA_PF_XCD"); return e && e[0] == '0'; }();
  const int64_t row = (int64_t)wgs_per_seq * num_seqs;
  return !off && row % (8 * wgs_per_group) == 0;
}

// Which form runs: 0 = by shape, 1 = the 256-thread form (one 128-query tile of one head per
// workgroup, two workgroups per CU), 2 = the 512-thread form.  VLA_PF_FORM sets the initial value.
// By shape: the 512-thread form halves the K/V bytes read per query and hides the softmax behind
// the next tile's first product, but its one workgroup per CU has nobody to cover its start-up
// and wind-down, so it takes the long prompts: from 3072 tokens where it pairs two heads of a KV
// group (measured 3-4 % ahead there, 5-25 % behind below 2048), from 512 tokens where it pairs
// query tiles (no grouping or an odd group size: 3-20 % ahead), and never at head sizes up to 64,
// whose tiles are too short on matrix work for the interleaving to pay (15-30 % behind).
std::atomic<int> g_pf_form{[] { const char* e = getenv("VLA_PF_FORM"); return e ? atoi(e) : 0; }()};

bool pf_use_il(int max_len, int head_size, int group_size) {
  const int f = g_pf_form.load(std::memory_order_relaxed);
  if (f) return f == 2;
  if (head_size <= 64) return false;
  return max_len >= (group_size % 2 == 0 ? 3072 : 512);
}

template <bool PREFIX, bool FP8, bool WIN>
void pf_launch(PrefillParams p, int head_size, int dtype, int num_seqs, int max_len, hipStream_t s) {
  const int gsz = p.num_heads / p.num_kv_heads;
  const bool il = pf_use_il(max_len, head_size, gsz);
  dim3 grid;
  if (il) {
    p.pair_heads = gsz % 2 == 0;
    const int hx = p.pair_heads ? 2 : 1;
    p.xcd_groups = pf_xcd_groups(p.num_heads / hx, gsz / hx, num_seqs);
    const int bm = p.pair_heads ? PF_BM : 2 * PF_BM;
    grid = dim3(p.num_heads / hx, num_seqs, (max_len + bm - 1) / bm);
  } else {
    // query runs of at most 64 / 32 tokens: two / four heads of a KV group per 128-row tile
    const int hpt = (max_len <= PF_BM / 4 && gsz % 4 == 0) ? 4 : (max_len <= PF_BM / 2 && gsz % 2 == 0) ? 2 : 1;
    p.heads_per_tile = hpt;
    p.xcd_groups = pf_xcd_groups(p.num_heads / hpt, gsz / hpt, num_seqs);
    const int rows = PF_BM / hpt;
    grid = dim3(p.num_heads / hpt, num_seqs, (max_len + rows - 1) / rows);
  }
#define VLA_PF(ST, DD)                                                                       \
  do {                                                                                       \
    if (il)                                                                                  \
      hipLaunchKernelGGL((prefill_attention_il_kernel<ST, DD, PREFIX, FP8, WIN>), grid,    \
                         dim3(PF2_THREADS), 0, s, p);                                      \
    else                                                                                     \
      hipLaunchKernelGGL((prefill_attention_kernel<ST, DD, PREFIX, FP8, WIN>), grid,       \
                         dim3(PF_THREADS), 0, s, p);                                       \
  } while (0)
#define VLA_PF_D(ST)                     \
  switch (head_size) {                   \
    case 24:                             \
      if constexpr (!FP8) VLA_PF(ST, 24); \
      break;                             \
    case 64: VLA_PF(ST, 64); break;      \
    case 80: VLA_PF(ST, 80); break;      \
    case 96: VLA_PF(ST, 96); break;      \
    case 112: VLA_PF(ST, 112); break;    \
    default: VLA_PF(ST, 128); break;     \
  }
#ifdef VLA_PF_QUICK   // one instantiation only: quick compiles while studying the kernel
  if constexpr (!PREFIX) VLA_PF(F16, 128);
#else
  if (dtype == VLA_F16) { VLA_PF_D(F16) } else { VLA_PF_D(BF16) }
#endif
#undef VLA_PF_D
#undef VLA_PF
}

}  // namespace

extern "C" int vla_set_prefill_form(int mode) {
  if (mode < 0 || mode > 2) return VLA_ERR_ARG;
  g_pf_form.store(mode, std::memory_order_relaxed);
  return VLA_OK;
}

extern "C" int vla_prefill_attention(void* out, const void* query, const void* key,
                                     const void* value, const int* cu_seqlens, int num_seqs,
                                     int max_seqlen, int num_heads, int num_kv_heads, int head_size,
                                     int64_t query_stride, int64_t key_stride, int64_t value_stride,
                                     int64_t out_stride, float scale, const float* alibi_slopes,
                                     int dtype, void* stream) {
  (void)hipGetLastError();
  if (num_seqs == 0 || max_seqlen == 0) return VLA_OK;
  if (!out || !query || !key || !value || !cu_seqlens || num_seqs < 0 || max_seqlen < 0 ||
      num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads)
    return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16) return VLA_ERR_DTYPE;
  if (!pf_head_size_ok(head_size)) return VLA_ERR_HEAD_SIZE;
  if ((query_stride | key_stride | value_stride | out_stride) % 8 || ((uintptr_t)query % 16) ||
      ((uintptr_t)key % 16) || ((uintptr_t)value % 16))
    return VLA_ERR_ARG;
  if (num_seqs > 65535 || (max_seqlen + PF_BM - 1) / PF_BM > 65535) return VLA_ERR_ARG;  // grid y, z
  if (key_stride >= (1 << 24) || value_stride >= (1 << 24)) return VLA_ERR_ARG;  // 32-bit piece offsets
  PrefillParams p{};
  p.out = out; p.q = query; p.k = key; p.v = value;
  p.cu_seqlens = cu_seqlens; p.alibi_slopes = alibi_slopes;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.q_stride = query_stride; p.k_stride = key_stride; p.v_stride = value_stride;
  p.out_stride = out_stride; p.scale = scale; p.sliding_window = 0;
  pf_launch<false, false, false>(p, head_size, dtype, num_seqs, max_seqlen, (hipStream_t)stream);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_prefix_prefill_attention(
    void* out, const void* query, const void* key, const void* value, const void* key_cache,
    const void* value_cache, const int* block_tables, int max_blocks_per_seq,
    const int* query_start_loc, const int* seq_lens, const int* context_lens, int num_seqs,
    int max_query_len, int num_heads, int num_kv_heads, int head_size, int block_size,
    int64_t query_stride, int64_t key_stride, int64_t value_stride, int64_t out_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, float scale, const float* alibi_slopes,
    int sliding_window, int dtype, int kv_dtype, float kv_scale, void* stream) {
  (void)hipGetLastError();
  if (num_seqs == 0 || max_query_len == 0) return VLA_OK;
  if (!out || !query || !key || !value || !key_cache || !value_cache || !block_tables ||
      !query_start_loc || !seq_lens || !context_lens || num_seqs < 0 || max_query_len < 0 ||
      max_blocks_per_seq <= 0 || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads)
    return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16) return VLA_ERR_DTYPE;
  if (kv_dtype != VLA_KV_AUTO && kv_dtype != VLA_KV_FP8_E4M3) return VLA_ERR_KV_DTYPE;
  if (!pf_head_size_ok(head_size)) return VLA_ERR_HEAD_SIZE;
  const bool fp8 = kv_dtype == VLA_KV_FP8_E4M3;
  if (fp8 && head_size % 16) return VLA_ERR_HEAD_SIZE;   // no [D/16][BS][16] K layout exists
  // a 64-key tile is a whole number of blocks, and a 16-byte piece of a value row stays in a block
  if (!(block_size == 16 || block_size == 32 || block_size == 64 || (!fp8 && block_size == 8)))
    return VLA_ERR_BLOCK_SIZE;
  const int eb = fp8 ? 1 : 2;
  if ((query_stride | key_stride | value_stride | out_stride) % 8 || ((uintptr_t)query % 16) ||
      ((uintptr_t)key % 16) || ((uintptr_t)value % 16) || ((uintptr_t)key_cache % 16) ||
      ((uintptr_t)value_cache % 16) || (kv_block_stride * eb) % 16 || (kv_head_stride * eb) % 16)
    return VLA_ERR_ARG;
  if (num_seqs > 65535 || (max_query_len + PF_BM - 1) / PF_BM > 65535) return VLA_ERR_ARG;  // grid y, z
  if (key_stride >= (1 << 24) || value_stride >= (1 << 24)) return VLA_ERR_ARG;  // 32-bit piece offsets
  PrefillParams p{};
  p.out = out; p.q = query; p.k = key; p.v = value;
  p.cu_seqlens = query_start_loc; p.alibi_slopes = alibi_slopes;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.q_stride = query_stride; p.k_stride = key_stride; p.v_stride = value_stride;
  p.out_stride = out_stride; p.scale = scale;
  p.sliding_window = sliding_window > 0 ? sliding_window : 0;
  p.k_cache = key_cache; p.v_cache = value_cache; p.block_tables = block_tables;
  p.seq_lens = seq_lens; p.ctx_lens = context_lens; p.max_blocks_per_seq = max_blocks_per_seq;
  p.log2_block_size = block_size == 8 ? 3 : block_size == 16 ? 4 : block_size == 32 ? 5 : 6;
  p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride; p.kv_scale = kv_scale;
  hipStream_t s = (hipStream_t)stream;
  if (p.sliding_window > 0) {
    if (fp8) pf_launch<true, true, true>(p, head_size, dtype, num_seqs, max_query_len, s);
    else pf_launch<true, false, true>(p, head_size, dtype, num_seqs, max_query_len, s);
  } else {
    if (fp8) pf_launch<true, true, false>(p, head_size, dtype, num_seqs, max_query_len, s);
    else pf_launch<true, false, false>(p, head_size, dtype, num_seqs, max_query_len, s);
  }
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}
