// Single-query paged attention for gfx950, 16-bit models (fp16 / bf16), KV cache in the model
// dtype or fp8 e4m3.  This is the throughput kernel of the backend.
//
// Work split.  One 256-thread workgroup per (sequence, KV head[, 512-token partition]) serves
// ALL query heads of that KV head (up to 16 per workgroup), so every K/V byte is fetched from
// HBM once per KV head and not once per query head as in the reference's grid
// (csrc/attention/attention_kernels.cu:734).  Each of the 4 waves walks its own share of the
// token range in steps (64 tokens for fp8 KV, 32 for 16-bit KV) with a private online softmax;
// the waves are merged through LDS at the end.
//
// Data path.  K and V go straight from HBM to registers with 16-byte loads per lane, one step
// ahead of their use (K of step i+1 is in flight during the softmax / PV of step i, V of step i
// during its QK); they are streamed once and shared by nothing, so staging them in LDS would
// only add a round trip.  With several query heads per KV head both products are small dense
// contractions, evaluated on the matrix cores as 16x16x32 tiles (rows = 16 tokens for q.K,
// rows = up to 16 heads for P.V): that keeps the vector ALU for the fp8 -> fp16 conversions
// and the softmax, which is what makes the kernel HBM-bound instead of issue-bound.  The only
// LDS traffic in the loop is the 16 x step probability tile, which has to change lanes between
// the two products.
//
//   q.K tile:  A = K   lane(g, r): token r of the 16-token block, 8 head dims
//              B = q^T lane(g, c): head c, the same 8 head dims
//              C       lane(g, c): head c, tokens 4g..4g+3
//   P.V tile:  A = P   lane(g, r): head r, 8 tokens          (read from LDS)
//              B = V   lane(g, c): head dim 16n + c, the same 8 tokens
//              C       lane(g, c): head dim 16n + c, heads 4g..4g+3
// The contraction index of an MFMA may be permuted freely as long as A and B agree, so each
// lane's 16-byte load is used as it arrives: an fp8 K load (16 head dims of one token) feeds
// two MFMAs, an fp8 V load (16 tokens of one head dim) likewise.
//
// Numerics: fp32 logits, fp32 running max / sum, probabilities rounded to the model dtype
// before the value product (as the reference does, attention_kernels.cu:398-400), fp32
// accumulation, normaliser 1 / (sum + 1e-6) (:342).  fp8 bytes convert exactly to fp16/bf16;
// kv_scale is folded into the logit scale and the output instead of being multiplied into
// every element (identical when kv_scale == 1).
#include <type_traits>

#include "vla_common.h"



#include <math.h>

namespace vla {
namespace {

// K and V are read once per launch, by exactly one workgroup (the whole GQA group of a KV head
// shares it).  With the nontemporal hint on those loads a launch over batch 256 x 4096 tokens of
// fp8 cache takes 0.335 ms instead of 0.370 (6.4 against 5.8 TB/s); a cache small enough to stay
// resident is faster without it, so the launch chooses (AttnParams::kv_nontemporal).
template <bool NT>
__device__ __forceinline__ u32x4 kv_load(const unsigned char* p) {
  if constexpr (NT)
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  else
    return *reinterpret_cast<const u32x4*>(p);
}


template <bool NT>
__device__ __forceinline__ u32x2 kv_load8(const unsigned char* p) {
  if constexpr (NT)
    return __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
  else
    return *reinterpret_cast<const u32x2*>(p);
}

constexpr int kThreads = 256;
constexpr int kWaves = 4;

template <class S>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<F16>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<BF16>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <class S>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  typename S::T x = S::from_f32(a), y = S::from_f32(b);
  return (uint32_t)__builtin_bit_cast(uint16_t, x) |
         ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}

// 8 fp8 bytes (two dwords) -> one MFMA operand of 8 16-bit floats
template <class S>
__device__ __forceinline__ u32x4 cvt8(uint32_t w0, uint32_t w1) {
  u32x2 a = fp8x4_to_16<S>(w0), b = fp8x4_to_16<S>(w1);
  u32x4 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
  return r;
}

template <class S>
__device__ __forceinline__ float pa_rnd(float v) { return S::to_f32(S::from_f32(v)); }

// NeoX rotation of one (x, y) = (dim d, dim d + D/2) pair with the rounding steps of the rotary
// kernel (csrc/pos_encoding_kernels.cu:10-36 in the model dtype: every product and the sum rounded)
template <class S>
__device__ __forceinline__ void rope_pair(float x, float y, float c, float s, float& xr, float& yr) {
#pragma clang fp contract(off)
  xr = pa_rnd<S>(pa_rnd<S>(x * c) - pa_rnd<S>(y * s));
  yr = pa_rnd<S>(pa_rnd<S>(y * c) + pa_rnd<S>(x * s));
}

// eight consecutive elements of a token's row as the model dtype would hold them: from the tensor,
// or from the split-K slabs (summed in slab order, then rounded)
template <class S>
__device__ __forceinline__ void row8(const typename S::T* t, const float* srow, int num_slabs,
                                     int64_t slab_stride, int col, float (&f)[8]) {
  if (srow) {
    f32x4 a = *reinterpret_cast<const f32x4*>(srow + col);
    f32x4 b = *reinterpret_cast<const f32x4*>(srow + col + 4);
    for (int p = 1; p < num_slabs; ++p) {
      a += *reinterpret_cast<const f32x4*>(srow + (int64_t)p * slab_stride + col);
      b += *reinterpret_cast<const f32x4*>(srow + (int64_t)p * slab_stride + col + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[e] = pa_rnd<S>(a[e]);
      f[4 + e] = pa_rnd<S>(b[e]);
    }
  } else {
    const u32x4 w = *reinterpret_cast<const u32x4*>(t);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      f[e] = S::to_f32(__builtin_bit_cast(typename S::T, (uint16_t)(w[e >> 1] >> (16 * (e & 1)))));
  }
}

// keep the first `nvalid` tokens of a 16-byte V load, zero the rest (they may hold NaNs,
// attention_kernels.cu:420-430)
template <bool FP8>
__device__ __forceinline__ u32x4 mask_tokens(u32x4 v, int nvalid) {
  constexpr int per_dword = FP8 ? 4 : 2;
  constexpr int bits = FP8 ? 8 : 16;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    int nv = nvalid - w * per_dword;
    nv = nv < 0 ? 0 : (nv > per_dword ? per_dword : nv);
    uint32_t m = nv == per_dword ? 0xffffffffu : ((1u << (bits * nv)) - 1u);
    v[w] &= m;
  }
  return v;
}

// D: head size (a multiple of 16; the q.K contraction is padded with zero query dims up to the
// next multiple of 4 K chunks), BS: KV block size (8, 16 or 32).  With 8-token blocks a 16-token
// tile spans two blocks: a lane's K load picks its block by its token (lanes 0-7 / 8-15 of a
// 16-lane group), a V row of a byte cache is 8 bytes, so a lane's 16 tokens of one head dim are
// two 8-byte loads from two blocks; a 16-bit V row is exactly one 16-byte load.
// KVNT: nontemporal K / V loads (a template parameter: behind a run-time branch the compiler
// merges the two otherwise identical load groups and drops the hint)
// FUSE: the launch also rotates this step's query and, with RopeFuse::write_kv, rotates the key
// and writes key and value into the cache (see RopeFuse; D == 128, the whole GQA group in one
// workgroup).  After the first K requests have gone out the workgroup rotates its heads' query
// once, a rotation pair per thread, into LDS, from where every wave takes its fragments: measured
// free (batch 256 x 4096 tokens, fp8 cache: 345 us against 360 for the rotary launch + the plain
// kernel).  The new token belongs to the LAST step of its sequence, which one wave owns: with
// write_kv that wave rotates the key, stores key and value into the cache and waits for the
// stores' acknowledgement once, before its K loop.  No other wave, and no other workgroup, reads
// those bytes in this launch (a step covers whole blocks, and a (block, KV head) region belongs to
// one workgroup).  The K loop is the unfused one, untouched.  The cache write is NOT free once the
// launch saturates HBM: 2048 workgroups' scattered byte stores inside the read stream cost 30 us
// at that size (375 us), more than the launch they save, so large batches keep the cache write in
// its own launch (key and value only) and small ones (a few sequences) take everything here.
template <class S, bool FP8, int D, int BS, bool PARTITIONED, bool KVNT, bool FUSE = false>
__global__ __launch_bounds__(kThreads, 2) void pa_mfma_kernel(AttnParams p, int heads_per_wg,
                                                              RopeFuse rf, int q_lds_off) {
  using T = typename S::T;
  constexpr int kBS = BS;
  constexpr int X = FP8 ? 16 : 8;          // elements per 16-byte K chunk
  constexpr int TG = FP8 ? 16 : 8;         // tokens per 16-byte V load
  constexpr int STEP = 4 * TG;             // tokens per wave step
  constexpr int NT = STEP / 16;            // q.K tiles (16 tokens each) per step
  constexpr int KL = (D + 4 * X - 1) / (4 * X);  // K loads per tile per lane
  constexpr int KS = FP8 ? 2 * KL : KL;    // MFMA k-steps per q.K tile
  constexpr int ND = D / 16;               // head-dim tiles of the output
  constexpr int PM = FP8 ? 2 : 1;          // MFMAs fed by one V load
  constexpr int CE = FP8 ? 1 : 2;          // cache element bytes
  constexpr int P_PITCH = STEP * 2 + 16;   // bytes per head row of the probability tile
  static_assert(D % 16 == 0 && (BS == 8 || BS == 16 || BS == 32), "head size multiple of 16, block 8/16/32");
  constexpr int NB = BS == 8 ? STEP / 8 : NT;   // block-table entries a step looks up

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;

  const int qpk = p.num_heads / p.num_kv_heads;
  const int groups_per_kv = (qpk + heads_per_wg - 1) / heads_per_wg;
  const int kv_head = blockIdx.x / groups_per_kv;
  const int head_base = kv_head * qpk + (blockIdx.x % groups_per_kv) * heads_per_wg;
  const int nheads = min(heads_per_wg, kv_head * qpk + qpk - head_base);
  const int seq = p.seq_order ? p.seq_order[blockIdx.y] : blockIdx.y, part = blockIdx.z;

  const int seq_len = p.seq_lens[seq];
  const int tok_begin = PARTITIONED ? part * VLA_PARTITION_SIZE : 0;
  if (tok_begin >= seq_len) {
    // V1 on an empty sequence (a padding row of a captured batch): the reference kernel ends
    // with exp_sum 0 and zero accumulators and stores zeros (attention_kernels.cu:333-345,
    // :483-495).  V2 partitions past the end store nothing (:116-119); its reduce writes zeros.
    if constexpr (!PARTITIONED) {
      T* out = reinterpret_cast<T*>(p.out) + ((int64_t)seq * p.num_heads + head_base) * D;
      for (int idx = threadIdx.x; idx < nheads * D; idx += kThreads) out[idx] = S::from_f32(0.f);
    }
    return;
  }
  const int tok_end = PARTITIONED ? min(tok_begin + VLA_PARTITION_SIZE, seq_len) : seq_len;
  const int last_blk = (tok_end - 1) / kBS;
  const int* bt = p.block_tables + (int64_t)seq * p.max_num_blocks_per_seq;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* p_lds = smem + wave * (16 * P_PITCH);  // this wave's probability tile

  // ---- query fragments (B operand of q.K), zero rows for heads this workgroup does not own
  u32x4 qf[KS];
  if constexpr (!FUSE) {
    const T* q = reinterpret_cast<const T*>(p.q) + (int64_t)seq * p.q_stride +
                 (int64_t)(head_base + c) * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = FP8 ? 16 * (g + 4 * (ks >> 1)) + 8 * (ks & 1) : 8 * (g + 4 * ks);
      // dims past the head (padding of the contraction) and heads this workgroup does not own
      qf[ks] = (c < nheads && d0 < D) ? *reinterpret_cast<const u32x4*>(q + d0)
                                       : u32x4{0, 0, 0, 0};
    }
  }
  const float alibi = (p.alibi_slopes && c < nheads) ? p.alibi_slopes[head_base + c] : 0.f;
  const float qk_scale = FP8 ? p.scale * p.kv_scale : p.scale;

  const unsigned char* kbase = reinterpret_cast<const unsigned char*>(p.k_cache) +
                               (int64_t)kv_head * p.kv_head_stride * CE;
  const unsigned char* vbase = reinterpret_cast<const unsigned char*>(p.v_cache) +
                               (int64_t)kv_head * p.kv_head_stride * CE;
  const int64_t blk_bytes = p.kv_block_stride * CE;
  // per-lane byte offsets inside a block; a 16-token tile t starts at token (16 t) % BS of its
  // block, and a lane's V load covers tile g (fp8: 16 tokens) or half of tile g >> 1 (8 tokens)
  const int k_lane_off = (g * kBS + (BS == 8 ? (c & 7) : c)) * 16;  // chunk g (+4j), token c (+ tile offset)
  const int v_tok_off = BS == 8 ? 0 : FP8 ? (16 * g) % kBS : (16 * (g >> 1)) % kBS + 8 * (g & 1);
  const int v_lane_off = (c * kBS + v_tok_off) * CE;
  // K chunk g + 4j lies past the head for the last j of some lanes: those lanes re-read chunk g
  // (finite data) against zero query dims
  int k_j_off[KL];
#pragma unroll
  for (int j = 0; j < KL; ++j) k_j_off[j] = ((g + 4 * j) * X < D) ? j * (4 * kBS * 16) : 0;

  f32x4 o[ND];
#pragma unroll
  for (int n = 0; n < ND; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int nsteps = (tok_end - tok_begin + STEP - 1) / STEP;

  // block numbers of a step, clamped to the last block that holds live tokens
  auto step_blocks = [&](int step, int (&blk)[NB]) {
    const int b0 = (tok_begin + step * STEP) / kBS;
#pragma unroll
    for (int t = 0; t < NB; ++t)
      blk[t] = bt[min(b0 + (BS == 8 ? t : (16 * t) / kBS), last_blk)];
  };
  auto load_k = [&](const int (&blk)[NB], u32x4 (&kb)[NT * KL]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int kblk;
      if constexpr (BS == 8) kblk = (c & 8) ? blk[2 * t + 1] : blk[2 * t];
      else kblk = blk[t];
      const unsigned char* kp = kbase + (int64_t)kblk * blk_bytes + k_lane_off +
                                ((16 * t) % kBS) * 16;
#pragma unroll
      for (int j = 0; j < KL; ++j) kb[t * KL + j] = kv_load<KVNT>(kp + k_j_off[j]);
    }
  };

  // ---- FUSE: the owner of the last step puts the new token's key and value into the cache
  bool kv_writer = false;
  if constexpr (FUSE)
    kv_writer = rf.write_kv && tok_end == seq_len && wave == (nsteps - 1) % kWaves &&
                rf.slots[seq] >= 0;
  auto write_new_kv = [&]() {
    if constexpr (FUSE) {
      const int64_t slot = rf.slots[seq];
      const int64_t blk = slot / kBS;
      const int off = (int)(slot - blk * kBS);
      const T* cs = reinterpret_cast<const T*>(rf.cos_sin) + rf.positions[seq] * D;
      const float* srow = rf.slabs ? rf.slabs + (int64_t)seq * rf.slab_row : nullptr;
      auto elem = [&](const void* t, int64_t stride, int col, int d) -> float {
        if (srow) {
          float a = srow[col + kv_head * D + d];
          for (int q = 1; q < rf.num_slabs; ++q)
            a += srow[(int64_t)q * rf.slab_stride + col + kv_head * D + d];
          return pa_rnd<S>(a);
        }
        return S::to_f32(reinterpret_cast<const T*>(t)[(int64_t)seq * stride + kv_head * D + d]);
      };
      unsigned char* kdst = const_cast<unsigned char*>(kbase) + blk * blk_bytes;
      unsigned char* vdst = const_cast<unsigned char*>(vbase) + blk * blk_bytes;
      auto put_k = [&](int d, float val) {
        unsigned char* a = kdst + ((d / X) * kBS + off) * 16 + (d % X) * CE;
        if constexpr (FP8) *a = fp8_e4m3_encode(val / p.kv_scale);
        else *reinterpret_cast<T*>(a) = S::from_f32(val);
      };
      auto put_v = [&](int d, float val) {
        unsigned char* a = vdst + (d * kBS + off) * CE;
        if constexpr (FP8) *a = fp8_e4m3_encode(val / p.kv_scale);
        else *reinterpret_cast<T*>(a) = S::from_f32(val);
      };
      // lane l: rotation pair (l, l + 64) of the key, dims l and l + 64 of the value
      float xr, yr;
      rope_pair<S>(elem(rf.k, rf.k_stride, rf.k_col, lane), elem(rf.k, rf.k_stride, rf.k_col, lane + 64),
                   S::to_f32(cs[lane]), S::to_f32(cs[64 + lane]), xr, yr);
      put_k(lane, xr);
      put_k(lane + 64, yr);
      put_v(lane, elem(rf.v, rf.v_stride, rf.v_col, lane));
      put_v(lane + 64, elem(rf.v, rf.v_stride, rf.v_col, lane + 64));
    }
  };
  // The first K requests go out BEFORE the rotation work, which then runs while they are in
  // flight -- except in the wave whose first step is already the one that holds the new token.
  const bool kv_first = FUSE && kv_writer && wave == nsteps - 1;
  if (kv_first) {
    write_new_kv();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have reached memory
  }
  int cur_blk[NB];
  u32x4 kb[NT * KL];
  if (wave < nsteps) {
    step_blocks(wave, cur_blk);
    load_k(cur_blk, kb);
  }
  if constexpr (FUSE) {
    // ---- the rotated query of the workgroup's heads: one rotation pair per thread into LDS, from
    // where every wave takes its fragments; and, in the owner of the last step, the cache write
    static_assert(D == 128, "whole-head NeoX rotation of a 128-dim head");
    constexpr int Q_PITCH = D * 2 + 16;          // bytes per head row
    unsigned char* q_lds = smem + q_lds_off;
    {
      const T* cs = reinterpret_cast<const T*>(rf.cos_sin) + rf.positions[seq] * D;
      const float* srow = rf.slabs ? rf.slabs + (int64_t)seq * rf.slab_row : nullptr;
      const T* q = reinterpret_cast<const T*>(p.q) + (int64_t)seq * p.q_stride + (int64_t)head_base * D;
      auto elem = [&](int i) -> float {          // element i of the workgroup's heads, unrotated
        if (srow) {
          const int col = rf.q_col + head_base * D + i;
          float a = srow[col];
          for (int t = 1; t < rf.num_slabs; ++t) a += srow[(int64_t)t * rf.slab_stride + col];
          return pa_rnd<S>(a);
        }
        return S::to_f32(q[i]);
      };
      for (int idx = threadIdx.x; idx < nheads * (D / 2); idx += kThreads) {
        const int h = idx / (D / 2), o = idx - h * (D / 2);
        float xr, yr;
        rope_pair<S>(elem(h * D + o), elem(h * D + D / 2 + o), S::to_f32(cs[o]),
                     S::to_f32(cs[D / 2 + o]), xr, yr);
        *reinterpret_cast<T*>(q_lds + h * Q_PITCH + o * 2) = S::from_f32(xr);
        *reinterpret_cast<T*>(q_lds + h * Q_PITCH + (D / 2 + o) * 2) = S::from_f32(yr);
      }
    }
    if (kv_writer && !kv_first) write_new_kv();
    __syncthreads();
    // the writer's stores have reached memory before it requests anything else: the step that
    // reads them back is one of its own, later ones
    if (kv_writer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = FP8 ? 16 * (g + 4 * (ks >> 1)) + 8 * (ks & 1) : 8 * (g + 4 * ks);
      qf[ks] = c < nheads ? *reinterpret_cast<const u32x4*>(q_lds + c * Q_PITCH + d0 * 2)
                          : u32x4{0, 0, 0, 0};
    }
  }

  for (int step = wave; step < nsteps; step += kWaves) {
    const int step_base = tok_begin + step * STEP;
    const bool tail = step_base + STEP > tok_end;  // wave-uniform

    // block numbers of this wave's next step (used after q.K)
    int nxt_blk[NB] = {};
    const bool has_next = step + kWaves < nsteps;
    if (has_next) step_blocks(step + kWaves, nxt_blk);

    // ---- V loads of this step (consumed after the softmax)
    u32x4 vb[ND];
    if constexpr (BS == 8 && FP8) {
      // tokens 16 g .. + 7 in block 2 g of the step, + 8 .. + 15 in block 2 g + 1
      int vlo = cur_blk[0], vhi = cur_blk[1];
#pragma unroll
      for (int t = 1; t < NT; ++t) {
        vlo = (g == t) ? cur_blk[2 * t] : vlo;
        vhi = (g == t) ? cur_blk[2 * t + 1] : vhi;
      }
      const unsigned char* vp0 = vbase + (int64_t)vlo * blk_bytes + v_lane_off;
      const unsigned char* vp1 = vbase + (int64_t)vhi * blk_bytes + v_lane_off;
#pragma unroll
      for (int n = 0; n < ND; ++n) {
        const u32x2 a = kv_load8<KVNT>(vp0 + n * (16 * kBS)), b = kv_load8<KVNT>(vp1 + n * (16 * kBS));
        vb[n] = u32x4{a[0], a[1], b[0], b[1]};
      }
    } else {
      int vblk;
      if constexpr (BS == 8) {           // 16-bit cache: tokens 8 g .. + 7 of the step are block g
        vblk = cur_blk[0];
#pragma unroll
        for (int t = 1; t < NB; ++t) vblk = (g == t) ? cur_blk[t] : vblk;
      } else if constexpr (FP8) {
        vblk = cur_blk[0];
#pragma unroll
        for (int t = 1; t < NT; ++t) vblk = (g == t) ? cur_blk[t] : vblk;
      } else {
        vblk = (g >> 1) ? cur_blk[1] : cur_blk[0];
      }
      const unsigned char* vp = vbase + (int64_t)vblk * blk_bytes + v_lane_off;
#pragma unroll
      for (int n = 0; n < ND; ++n) vb[n] = kv_load<KVNT>(vp + n * (16 * kBS * CE));
    }

    // ---- logits: NT tiles of 16 tokens x 16 heads
    f32x4 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KL; ++j) {
        const u32x4 kk = kb[t * KL + j];
        if constexpr (FP8) {
          acc = mfma16<S>(cvt8<S>(kk[0], kk[1]), qf[2 * j], acc);
          acc = mfma16<S>(cvt8<S>(kk[2], kk[3]), qf[2 * j + 1], acc);
        } else {
          acc = mfma16<S>(kk, qf[j], acc);
        }
      }
      s[t] = acc;
    }

    // ---- K loads of the next step go out now; they land during softmax + P.V
    if (has_next) load_k(nxt_blk, kb);

    // ---- online softmax for head c over this lane's tokens
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tok = step_base + 16 * t + 4 * g + i;
        float v = s[t][i] * qk_scale;
        v += alibi * (float)(tok - seq_len + 1);
        v = (tok < tok_end) ? v : -INFINITY;
        s[t][i] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = __expf(s[t][i] - m_new);
        psum += e[i];
      }
      u32x2 w;
      w[0] = pack2<S>(e[0], e[1]);
      w[1] = pack2<S>(e[2], e[3]);
      *reinterpret_cast<u32x2*>(p_lds + c * P_PITCH + (16 * t + 4 * g) * 2) = w;
    }
    l_run = l_run * alpha + psum;

    // rescale the output rows (heads 4g..4g+3 live in this lane) only when some max moved
    if (__any(m_new > m_run)) {
      float a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = __shfl(alpha, 4 * g + i, 64);
#pragma unroll
      for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[n][i] *= a[i];
    }
    m_run = m_new;

    // ---- probability fragments (A operand of P.V): head c, this lane's 8 tokens per MFMA
    __builtin_amdgcn_wave_barrier();
    u32x4 pf[PM];
#pragma unroll
    for (int mm = 0; mm < PM; ++mm)
      pf[mm] = *reinterpret_cast<const u32x4*>(p_lds + c * P_PITCH + (TG * g + 8 * mm) * 2);
    __builtin_amdgcn_wave_barrier();

    // ---- P.V
    const int nvalid = tok_end - (step_base + TG * g);  // live tokens in this lane's V loads
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      u32x4 vv = vb[n];
      if (tail) vv = mask_tokens<FP8>(vv, nvalid);
      if constexpr (FP8) {
        o[n] = mfma16<S>(pf[0], cvt8<S>(vv[0], vv[1]), o[n]);
        o[n] = mfma16<S>(pf[1], cvt8<S>(vv[2], vv[3]), o[n]);
      } else {
        o[n] = mfma16<S>(pf[0], vv, o[n]);
      }
    }

#pragma unroll
    for (int t = 0; t < NB; ++t) cur_blk[t] = nxt_blk[t];
  }

  // ---- merge the waves
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  __syncthreads();  // every wave is done with its probability tile
  float* ml = reinterpret_cast<float*>(smem);            // [kWaves][2][16]  max, sum
  float* ol = ml + kWaves * 32;                          // [kWaves][heads_per_wg][D]
  if (g == 0) {
    ml[wave * 32 + c] = m_run;
    ml[wave * 32 + 16 + c] = l_run;
  }
#pragma unroll
  for (int n = 0; n < ND; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int h = 4 * g + i;
      if (h < nheads) ol[(wave * heads_per_wg + h) * D + 16 * n + c] = o[n][i];
    }
  __syncthreads();

  const float out_scale = FP8 ? p.kv_scale : 1.f;
  T* out = reinterpret_cast<T*>(p.out);
  for (int idx = threadIdx.x; idx < nheads * D; idx += kThreads) {
    const int h = idx / D, d = idx - h * D;
    float M = ml[h];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) M = fmaxf(M, ml[w * 32 + h]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const float f = __expf(ml[w * 32 + h] - M);
      L += ml[w * 32 + 16 + h] * f;
      O += ol[(w * heads_per_wg + h) * D + d] * f;
    }
    const float r = O * out_scale / (L + 1e-6f);
    const int64_t sh = (int64_t)seq * p.num_heads + head_base + h;
    if (PARTITIONED) {
      out[(sh * p.max_num_partitions + part) * D + d] = S::from_f32(r);
      if (d == 0) {
        p.exp_sums[sh * p.max_num_partitions + part] = L;
        p.max_logits[sh * p.max_num_partitions + part] = M;
      }
    } else {
      out[sh * D + d] = S::from_f32(r);
    }
  }
}

template <class S, bool FP8, int D, int BS>
int launch_b(const AttnParams& p, hipStream_t s, const RopeFuse* fuse) {
  const int qpk = p.num_heads / p.num_kv_heads;
  // query heads per workgroup: the whole GQA group when it fits the 16 rows of a tile
  int hpw = qpk <= 16 ? qpk : 16;
  const int groups_per_kv = (qpk + hpw - 1) / hpw;
  constexpr int STEP = FP8 ? 64 : 32;
  const size_t p_bytes = (size_t)kWaves * 16 * (STEP * 2 + 16);
  const size_t merge_bytes = (size_t)kWaves * 32 * 4 + (size_t)kWaves * hpw * D * 4;
  const size_t smem = p_bytes > merge_bytes ? p_bytes : merge_bytes;
  const size_t q_bytes = (size_t)hpw * (D * 2 + 16);   // FUSE: the rotated query of the workgroup
  if (smem + q_bytes > 64 * 1024) return -1;  // beyond the default dynamic-LDS limit: generic kernel
  dim3 grid(p.num_kv_heads * groups_per_kv, p.num_seqs,
            p.max_num_partitions > 0 ? p.max_num_partitions : 1);
#define VLA_PA_LAUNCH(PART, NTL, FUSED)                                                          \
  hipLaunchKernelGGL((pa_mfma_kernel<S, FP8, D, BS, PART, NTL, FUSED>), grid, dim3(kThreads),   \
                     smem + (FUSED ? q_bytes : 0), s, p, hpw, rf, (int)smem)
#define VLA_PA_NT(PART, FUSED)                                                                   \
  do {                                                                                           \
    if (p.kv_nontemporal) VLA_PA_LAUNCH(PART, true, FUSED); else VLA_PA_LAUNCH(PART, false, FUSED); \
  } while (0)
  const RopeFuse rf = fuse ? *fuse : RopeFuse{};
  if constexpr (D == 128) {
    if (fuse) {
      if (groups_per_kv != 1) return -1;
      if (p.max_num_partitions > 0) VLA_PA_NT(true, true); else VLA_PA_NT(false, true);
      VLA_HIP_CHECK(hipGetLastError());
      return VLA_OK;
    }
  }
  if (fuse) return -1;
  if (p.max_num_partitions > 0) VLA_PA_NT(true, false); else VLA_PA_NT(false, false);
#undef VLA_PA_NT
#undef VLA_PA_LAUNCH
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

template <class S, bool FP8, int D>
int launch_t(const AttnParams& p, hipStream_t s, const RopeFuse* fuse) {
  return p.block_size == 16 ? launch_b<S, FP8, D, 16>(p, s, fuse)
         : p.block_size == 32 ? launch_b<S, FP8, D, 32>(p, s, fuse) : launch_b<S, FP8, D, 8>(p, s, fuse);
}

template <class S, bool FP8>
int launch_d(const AttnParams& p, hipStream_t s, const RopeFuse* fuse) {
  switch (p.head_size) {
    case 64: return launch_t<S, FP8, 64>(p, s, fuse);
    case 80: return launch_t<S, FP8, 80>(p, s, fuse);
    case 96: return launch_t<S, FP8, 96>(p, s, fuse);
    case 112: return launch_t<S, FP8, 112>(p, s, fuse);
    case 128: return launch_t<S, FP8, 128>(p, s, fuse);
    case 192: return launch_t<S, FP8, 192>(p, s, fuse);
    case 256: return launch_t<S, FP8, 256>(p, s, fuse);
    default: return -1;
  }
}

}  // namespace

bool paged_attention_rope_fusable(int num_heads, int num_kv_heads, int head_size, int block_size,
                                  int dtype) {
  if (num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads) return false;
  return head_size == 128 && num_heads / num_kv_heads <= 16 &&
         (block_size == 8 || block_size == 16 || block_size == 32) &&
         (dtype == VLA_F16 || dtype == VLA_BF16);
}

int launch_paged_attention_mfma(const AttnParams& p, int dtype, int kv_dtype, hipStream_t s,
                                const RopeFuse* fuse) {
  if (p.block_size != 8 && p.block_size != 16 && p.block_size != 32) return -1;
  if (dtype != VLA_F16 && dtype != VLA_BF16) return -1;
  const bool fp8 = kv_dtype == VLA_KV_FP8_E4M3;
  const int ce = fp8 ? 1 : 2;
  // 16-byte loads: aligned bases and strides, 16-byte aligned query rows
  if (((uintptr_t)p.k_cache | (uintptr_t)p.v_cache) % 16) return -1;
  if ((p.kv_block_stride * ce) % 16 || (p.kv_head_stride * ce) % 16) return -1;
  if (fuse) {
    if (!paged_attention_rope_fusable(p.num_heads, p.num_kv_heads, p.head_size, p.block_size, dtype))
      return -1;
    if ((uintptr_t)fuse->cos_sin % 16) return -1;
    if (fuse->slabs) {
      if (((uintptr_t)fuse->slabs % 16) || fuse->slab_stride % 4 || fuse->slab_row % 4 ||
          fuse->q_col % 4 || fuse->k_col % 4 || fuse->v_col % 4 || fuse->num_slabs < 1)
        return -1;
    } else if (((uintptr_t)p.q % 16) || (p.q_stride * 2) % 16 || (fuse->write_kv && (!fuse->k || !fuse->v))) {
      return -1;
    }
  } else if (((uintptr_t)p.q % 16) || (p.q_stride * 2) % 16) {
    return -1;
  }
  if (dtype == VLA_F16)
    return fp8 ? launch_d<F16, true>(p, s, fuse) : launch_d<F16, false>(p, s, fuse);
  return fp8 ? launch_d<BF16, true>(p, s, fuse) : launch_d<BF16, false>(p, s, fuse);
}

}  // namespace vla
