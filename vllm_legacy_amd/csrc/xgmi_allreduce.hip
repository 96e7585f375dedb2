// SUM all-reduce of a tensor-parallel partial output by direct loads and stores between the
// GPUs of one xGMI node (no RCCL call on the path).
//
// Takes the place of the reference's custom all-reduce (csrc/custom_all_reduce.cuh:180-205
// barriers, :297-356 two-shot kernel, :442-449 dispatch; wrapper
// vllm/distributed/device_communicators/custom_all_reduce.py:78-303), which is CUDA-only
// (custom_all_reduce.py:28-31): on ROCm the reference falls back to RCCL for the 65 small
// all-reduces of a Llama decode step.
//
// Every rank owns one region of uncached device memory, exported over HIP IPC and mapped by
// all peers:
//   flags [MAX_BLOCKS][MAX_RANKS] u32   slot [b][p] is written by rank p's block b only
//   seq   [MAX_BLOCKS] u32              private to the owner: barriers passed so far, per block
//   err   u32                           private: barriers that gave up waiting
//   data  2 x max_bytes                 message staging, double buffered by call parity
// One kernel per call, the same grid on every rank, three steps per block:
//   1. copy this rank's input into its own staging half;            barrier
//   2. rank r sums slice r over all ranks' staging halves (fp32, rank order 0..N-1, so the bits
//      do not depend on who reads them) and writes it back in place; barrier
//   3. gather every slice from its owner's staging half into `out`.
// Block b touches the same element set on every rank in every step (a strided walk of each
// slice), so a barrier only has to pair block b of one rank with block b of the others: each
// block signals by storing its barrier count into its slot on every peer and waits until its
// own slots show that count.  The counts live in device memory and advance inside the kernel,
// so a captured graph replays correctly.  Double buffering makes a closing barrier unnecessary:
// a rank can only start writing a half again two calls later, and it passes the first barrier
// of the call in between only after every peer has finished reading in the call before.
// A wait gives up after a bounded TIME (wall clock, default 10 s: a peer that is still repacking
// weights or capturing a graph is merely late, not gone), so the kernel always terminates.  A
// timeout can not go unnoticed on any rank: the rank that gives up counts it in `err` of EVERY
// rank's region (its own and, by a system-scope atomic, each peer's).  `err` is sticky, a call
// reads it when it starts and again before it ends, and a call that finds it set stores NaN in
// its output instead of a sum -- so the late peer, which on arrival finds its flags already set
// and would sail through both barriers gathering slices that were reduced without its data, is
// poisoned just like the rank that waited for it (the barrier counts of the ranks may no longer
// agree, so nothing after a timeout can be trusted anywhere).
// Calls on one communicator must be issued in the same order with the same sizes by all ranks,
// on one stream per rank.
#include "vla_common.h"

#include <stdlib.h>
#include <string.h>

#include <new>

namespace vla {
namespace {

constexpr int AR_MAX_RANKS = 8;
constexpr int AR_MAX_BLOCKS = 96;
constexpr int AR_THREADS = 512;
constexpr int64_t AR_FLAG_OFF = 0;
constexpr int64_t AR_SEQ_OFF = (int64_t)AR_MAX_BLOCKS * AR_MAX_RANKS * 4;
constexpr int64_t AR_ERR_OFF = AR_SEQ_OFF + AR_MAX_BLOCKS * 4;
constexpr int64_t AR_DATA_OFF = 4096;
constexpr double AR_WAIT_SECONDS = 10.0;                 // default bound of one barrier wait
constexpr uint32_t AR_POISON = 0x7fff7fffu;              // NaN as two fp16, two bf16 or one fp32
static_assert(AR_ERR_OFF + 4 <= AR_DATA_OFF, "metadata fits its page");

struct ArParams {
  unsigned char* region[AR_MAX_RANKS];   // every rank's region as mapped here; [rank] is local
  int rank, world;
  int64_t max_bytes;
  uint64_t wait_ticks;                   // bound of one barrier wait in wall_clock64() ticks
};

struct ArContext {
  ArParams p;
  int num_blocks;
  int device;
};

// All threads of the block have issued the stores the peers are about to read.  Returns true
// (to every thread of the block) when some peer did not arrive within the time bound.  A rank
// that has already lost a barrier (`failed`) still signals its peers but no longer waits for
// them: its output is void anyway, and it should get out of the way quickly.
__device__ __forceinline__ bool ar_barrier(const ArParams& P, uint32_t count, bool failed) {
  __threadfence_system();
  __syncthreads();
  const int t = threadIdx.x;
  int timed_out = 0;
  if (t < P.world) {
    uint32_t* theirs = reinterpret_cast<uint32_t*>(P.region[t] + AR_FLAG_OFF) +
                       (blockIdx.x * AR_MAX_RANKS + P.rank);
    __hip_atomic_store(theirs, count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* mine = reinterpret_cast<uint32_t*>(P.region[P.rank] + AR_FLAG_OFF) +
                     (blockIdx.x * AR_MAX_RANKS + t);
    const uint64_t t0 = wall_clock64();
    unsigned polls = 0;
    // counts only grow; the signed difference survives a wrap of the counter
    while (!failed &&
           (int32_t)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) -
                     count) < 0) {
      if ((++polls & 1023u) == 0 && wall_clock64() - t0 > P.wait_ticks) {
        // tell every rank, the absent one included: its region is mapped here whether or not
        // its process is making progress
        for (int r = 0; r < P.world; ++r)
          __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(P.region[r] + AR_ERR_OFF), 1u,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  timed_out = __syncthreads_or(timed_out);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return failed || timed_out != 0;
}

// Whether some rank has lost a barrier so far (`err` is sticky; peers bump it over the link).
// One thread reads, the block agrees: other blocks of this launch, or a peer, may be adding to
// `err` at this moment, and threads that each read for themselves could disagree and poison only
// part of the block's output.
__device__ __forceinline__ bool ar_failed_before(const ArParams& P) {
  const uint32_t* err = reinterpret_cast<const uint32_t*>(P.region[P.rank] + AR_ERR_OFF);
  const int bad = threadIdx.x == 0 &&
                  __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
  return __syncthreads_or(bad) != 0;
}

template <class S>
__device__ __forceinline__ void ar_accumulate(float (&acc)[8], const u32x4 w) {
  using T = typename S::T;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint16_t lo = (uint16_t)(w[i] & 0xffffu), hi = (uint16_t)(w[i] >> 16);
    acc[2 * i] += S::to_f32(__builtin_bit_cast(T, lo));
    acc[2 * i + 1] += S::to_f32(__builtin_bit_cast(T, hi));
  }
}
template <>
__device__ __forceinline__ void ar_accumulate<F32>(float (&acc)[8], const u32x4 w) {
  const f32x4 f = __builtin_bit_cast(f32x4, w);
  acc[0] += f[0];
  acc[1] += f[1];
  acc[2] += f[2];
  acc[3] += f[3];
}

template <class S>
__device__ __forceinline__ u32x4 ar_pack(const float (&acc)[8]) {
  using T = typename S::T;
  u32x4 w;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const T a = S::from_f32(acc[2 * i]), b = S::from_f32(acc[2 * i + 1]);
    w[i] = (uint32_t)__builtin_bit_cast(uint16_t, a) |
           ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
  }
  return w;
}
template <>
__device__ __forceinline__ u32x4 ar_pack<F32>(const float (&acc)[8]) {
  const f32x4 f = {acc[0], acc[1], acc[2], acc[3]};
  return __builtin_bit_cast(u32x4, f);
}

// dst[i] = src[i] for i = first, first + stride, ... < len, four loads in flight per thread
__device__ __forceinline__ void ar_copy(u32x4* dst, const u32x4* src, int64_t first, int64_t len,
                                        int64_t stride) {
  int64_t i = first;
  for (; i + 3 * stride < len; i += 4 * stride) {
    const u32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a;
    dst[i + stride] = b;
    dst[i + 2 * stride] = c;
    dst[i + 3 * stride] = d;
  }
  for (; i < len; i += stride) dst[i] = src[i];
}

// dst[i] = S(sum over slabs of slab[p][i]) for 16-byte units i = first, first + stride, ... < len
// of dst (8 elements each): fp32 sums in slab order rounded once, the arithmetic of the int4
// GEMM's own split-K reduction
template <class S>
__device__ __forceinline__ void ar_copy_slabs(u32x4* dst, const float* slabs, int num_slabs,
                                              int64_t slab_stride, int64_t first, int64_t len,
                                              int64_t stride) {
  for (int64_t i = first; i < len; i += stride) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + 2 * i;
    f32x4 lo = src[0], hi = src[1];
    for (int p = 1; p < num_slabs; ++p) {
      const f32x4* q = reinterpret_cast<const f32x4*>(slabs + (int64_t)p * slab_stride) + 2 * i;
      lo += q[0];
      hi += q[1];
    }
    const float acc[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    dst[i] = ar_pack<S>(acc);
  }
}

// nvec: 16-byte units of the message; per: units per slice (the last slice may be short or empty).
// SLABS: the input is num_slabs fp32 partial products [num_slabs][numel] (an int4 GEMM whose K
// range was split) that are summed and rounded to S while they are published.
template <class S, bool SLABS = false>
__global__ __launch_bounds__(AR_THREADS) void xgmi_two_shot_kernel(ArParams P, u32x4* out,
                                                                   const u32x4* in, int64_t nvec,
                                                                   int64_t per, int num_slabs = 0,
                                                                   int64_t slab_stride = 0) {
  unsigned char* self = P.region[P.rank];
  uint32_t* seq_slot = reinterpret_cast<uint32_t*>(self + AR_SEQ_OFF) + blockIdx.x;
  const uint32_t seq = __hip_atomic_load(seq_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t half = AR_DATA_OFF + (int64_t)((seq >> 1) & 1u) * P.max_bytes;
  const int64_t stride = (int64_t)gridDim.x * AR_THREADS;
  const int64_t first = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x;

  // 1. publish the input
  u32x4* mine = reinterpret_cast<u32x4*>(self + half);
  for (int s = 0; s < P.world; ++s) {
    const int64_t base = s * per, len = min(per, nvec - base);
    if constexpr (SLABS)
      ar_copy_slabs<S>(mine + base, reinterpret_cast<const float*>(in) + 8 * base, num_slabs,
                       slab_stride, first, len, stride);
    else
      ar_copy(mine + base, in + base, first, len, stride);
  }
  bool failed = ar_failed_before(P);
  failed = ar_barrier(P, seq + 1, failed);

  // 2. reduce this rank's slice in place
  {
    const int64_t base = P.rank * per, len = min(per, nvec - base);
    for (int64_t i = first; i < len; i += stride) {
      // all ranks' loads in flight together: the link latency is paid once per element
      u32x4 w[AR_MAX_RANKS];
#pragma unroll
      for (int r = 0; r < AR_MAX_RANKS; ++r)
        if (r < P.world) w[r] = reinterpret_cast<const u32x4*>(P.region[r] + half)[base + i];
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < AR_MAX_RANKS; ++r)
        if (r < P.world) ar_accumulate<S>(acc, w[r]);
      mine[base + i] = ar_pack<S>(acc);
    }
  }
  failed = ar_barrier(P, seq + 2, failed);

  // 3. gather the slices from their owners
  for (int s = 0; s < P.world; ++s) {
    const int64_t base = s * per, len = min(per, nvec - base);
    const u32x4* src = reinterpret_cast<const u32x4*>(P.region[s] + half);
    ar_copy(out + base, src + base, first, len, stride);
  }
  failed = ar_failed_before(P) || failed;   // a peer may have given up on this rank meanwhile
  if (failed) {  // a lost barrier: this block's part of the output is not a sum -- make it NaN
    // (each thread overwrites exactly the elements it has just gathered: no store of another
    // thread can land after the poison)
    const u32x4 nan = {AR_POISON, AR_POISON, AR_POISON, AR_POISON};
    for (int s = 0; s < P.world; ++s) {
      const int64_t base = s * per, len = min(per, nvec - base);
      for (int64_t i = first; i < len; i += stride) out[base + i] = nan;
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(seq_slot, seq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class S>
__device__ __forceinline__ void ar_unpack(const u32x4 w, float (&f)[8]) {
  using T = typename S::T;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = S::to_f32(__builtin_bit_cast(T, (uint16_t)(w[i] & 0xffffu)));
    f[2 * i + 1] = S::to_f32(__builtin_bit_cast(T, (uint16_t)(w[i] >> 16)));
  }
}

template <class S>
__device__ __forceinline__ float ar_rnd(float v) { return S::to_f32(S::from_f32(v)); }

// The all-reduce followed by the layer's next step, residual = sum + residual and
// out = RMSNorm(residual) * weight (csrc/layernorm_kernels.cu:52-91, the pair
// vllm/model_executor/models/llama.py:213-224 runs after each row-parallel linear), in one launch.
// The element-to-block map is by rows instead of the strided walk above: block b owns rows
// b, b + G, ... in all three steps (so the pairing of block b with block b of the peers still
// covers every dependency), rank r reduces the r-th column range of each row, and in step 3 a
// block holds whole rows of the sum, which is what the normalisation needs.
// rv: 16-byte units per row (hidden / 8), a multiple of the world size, at most 2 * AR_THREADS.
constexpr int AR_ROW_CHUNK = 8;   // rows whose loads are in flight together in step 3
// PAIR (two ranks): step 2 and its barrier are dropped -- step 3 reads the row from both staging
// halves and sums it itself (rank order, the same bits), which moves the same number of bytes
// over the link as reduce + gather do.  The barrier count still advances by two per call, so
// calls of both kinds can follow each other on one communicator.
template <class S, bool SLABS, bool PAIR>
__global__ __launch_bounds__(AR_THREADS) void xgmi_two_shot_norm_kernel(
    ArParams P, u32x4* out, const u32x4* in, u32x4* residual, const u32x4* weight, float eps,
    int rows, int rv, int num_slabs, int64_t slab_stride) {
  __shared__ float red[AR_ROW_CHUNK][AR_THREADS / 64];
  unsigned char* self = P.region[P.rank];
  uint32_t* seq_slot = reinterpret_cast<uint32_t*>(self + AR_SEQ_OFF) + blockIdx.x;
  const uint32_t seq = __hip_atomic_load(seq_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t half = AR_DATA_OFF + (int64_t)((seq >> 1) & 1u) * P.max_bytes;
  const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  u32x4* mine = reinterpret_cast<u32x4*>(self + half);

  // 1. publish this block's rows
  for (int row = b; row < rows; row += G) {
    const int64_t r0 = (int64_t)row * rv;
    if constexpr (SLABS)
      ar_copy_slabs<S>(mine + r0, reinterpret_cast<const float*>(in) + 8 * r0, num_slabs,
                       slab_stride, t, rv, AR_THREADS);
    else
      for (int v = t; v < rv; v += AR_THREADS) mine[r0 + v] = in[r0 + v];
  }
  bool failed = ar_failed_before(P);
  failed = ar_barrier(P, PAIR ? seq + 2 : seq + 1, failed);

  // 2. this rank's column range of this block's rows, summed over the ranks in place
  const int cw = rv / P.world, c0 = P.rank * cw;
  if constexpr (!PAIR) {
    const int my_rows = b < rows ? (rows - b + G - 1) / G : 0;
    for (int item = t; item < my_rows * cw; item += AR_THREADS) {
      const int64_t idx = (int64_t)(b + (item / cw) * G) * rv + c0 + item % cw;
      u32x4 w[AR_MAX_RANKS];
#pragma unroll
      for (int r = 0; r < AR_MAX_RANKS; ++r)
        if (r < P.world) w[r] = reinterpret_cast<const u32x4*>(P.region[r] + half)[idx];
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < AR_MAX_RANKS; ++r)
        if (r < P.world) ar_accumulate<S>(acc, w[r]);
      mine[idx] = ar_pack<S>(acc);
    }
    failed = ar_barrier(P, seq + 2, failed);
  }

  // 3. whole rows of the sum from their owners; add the residual, normalise.  A chunk of rows
  // has all its loads in flight together and shares one reduction of its sums of squares.
  const float inv_hidden = 1.0f / (float)(rv * 8);
  const int lane = t & 63, wv = t >> 6;
  for (int row0 = b; row0 < rows; row0 += AR_ROW_CHUNK * G) {
    u32x4 sum_w[AR_ROW_CHUNK][2], res_w[AR_ROW_CHUNK][2];
#pragma unroll
    for (int c = 0; c < AR_ROW_CHUNK; ++c) {
      const int row = row0 + c * G;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = t + k * AR_THREADS;
        if (row < rows && v < rv) {
          const int64_t idx = (int64_t)row * rv + v;
          if constexpr (PAIR) {
            const u32x4 w0 = reinterpret_cast<const u32x4*>(P.region[0] + half)[idx];
            const u32x4 w1 = reinterpret_cast<const u32x4*>(P.region[1] + half)[idx];
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            ar_accumulate<S>(acc, w0);
            ar_accumulate<S>(acc, w1);
            sum_w[c][k] = ar_pack<S>(acc);
          } else {
            sum_w[c][k] = reinterpret_cast<const u32x4*>(P.region[v / cw] + half)[idx];
          }
          res_w[c][k] = residual[idx];
        }
      }
    }
    // x = sum + residual (kept as packed model-dtype words), per-row sums of squares
#pragma unroll
    for (int c = 0; c < AR_ROW_CHUNK; ++c) {
      const int row = row0 + c * G;
      float var = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = t + k * AR_THREADS;
        if (row < rows && v < rv) {
          float a[8], r[8], x[8];
          ar_unpack<S>(sum_w[c][k], a);
          ar_unpack<S>(res_w[c][k], r);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            x[e] = ar_rnd<S>(a[e] + r[e]);
            var += x[e] * x[e];
          }
          sum_w[c][k] = ar_pack<S>(x);
          residual[(int64_t)row * rv + v] = sum_w[c][k];
        }
      }
      var = wave_sum(var);
      if (lane == 0) red[c][wv] = var;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < AR_ROW_CHUNK; ++c) {
      const int row = row0 + c * G;
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < AR_THREADS / 64; ++i) var += red[c][i];
      const float inv = rsqrtf(var * inv_hidden + eps);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = t + k * AR_THREADS;
        if (row < rows && v < rv) {
          float x[8], w[8], y[8];
          ar_unpack<S>(sum_w[c][k], x);
          ar_unpack<S>(weight[v], w);
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = ar_rnd<S>(x[e] * inv) * w[e];
          out[(int64_t)row * rv + v] = ar_pack<S>(y);
        }
      }
    }
    __syncthreads();                                   // red is reused by the next chunk
  }
  failed = ar_failed_before(P) || failed;   // a peer may have given up on this rank meanwhile
  if (failed) {  // a lost barrier: the rows this block wrote are not what they claim to be
    const u32x4 nan = {AR_POISON, AR_POISON, AR_POISON, AR_POISON};
    for (int row = b; row < rows; row += G)
      for (int v = t; v < rv; v += AR_THREADS) {
        out[(int64_t)row * rv + v] = nan;
        residual[(int64_t)row * rv + v] = nan;
      }
  }
  if (threadIdx.x == 0) __hip_atomic_store(seq_slot, seq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace
}  // namespace vla

using namespace vla;

extern "C" int vla_ar_region_bytes(int64_t max_bytes, int64_t* out) {
  if (max_bytes <= 0 || max_bytes % 16 || !out) return VLA_ERR_ARG;
  *out = AR_DATA_OFF + 2 * max_bytes;
  return VLA_OK;
}

extern "C" int vla_ar_alloc_region(int64_t max_bytes, void** region) {
  int64_t bytes = 0;
  if (!region || vla_ar_region_bytes(max_bytes, &bytes) != VLA_OK) return VLA_ERR_ARG;
  void* p = nullptr;
  VLA_HIP_CHECK(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached));
  hipError_t e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p);
    VLA_HIP_CHECK(e);
  }
  *region = p;
  return VLA_OK;
}

extern "C" int vla_ar_free_region(void* region) {
  if (region) VLA_HIP_CHECK(hipFree(region));
  return VLA_OK;
}

extern "C" int vla_ar_export(void* region, unsigned char* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  if (!region || !handle64) return VLA_ERR_ARG;
  hipIpcMemHandle_t h;
  VLA_HIP_CHECK(hipIpcGetMemHandle(&h, region));
  memcpy(handle64, &h, sizeof(h));
  return VLA_OK;
}

extern "C" int vla_ar_import(const unsigned char* handle64, int peer_device, void** peer_region) {
  if (!handle64 || !peer_region) return VLA_ERR_ARG;
  int cur = 0;
  VLA_HIP_CHECK(hipGetDevice(&cur));
  if (peer_device >= 0 && peer_device != cur) {
    int can = 0;
    VLA_HIP_CHECK(hipDeviceCanAccessPeer(&can, cur, peer_device));
    if (!can) {
      set_last_hip_error((int)hipErrorPeerAccessUnsupported);
      return VLA_ERR_HIP;
    }
    hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
    else VLA_HIP_CHECK(e);
  }
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  void* p = nullptr;
  VLA_HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  *peer_region = p;
  return VLA_OK;
}

extern "C" int vla_ar_unimport(void* peer_region) {
  if (peer_region) VLA_HIP_CHECK(hipIpcCloseMemHandle(peer_region));
  return VLA_OK;
}

extern "C" int vla_ar_create(void** ctx, int rank, int world, void* const* regions,
                             int64_t max_bytes, int num_blocks) {
  if (!ctx || !regions || world < 2 || world > AR_MAX_RANKS || rank < 0 || rank >= world ||
      max_bytes <= 0 || max_bytes % 16 || num_blocks <= 0 || num_blocks > AR_MAX_BLOCKS)
    return VLA_ERR_ARG;
  for (int r = 0; r < world; ++r)
    if (!regions[r]) return VLA_ERR_ARG;
  ArContext* c = new (std::nothrow) ArContext();
  if (!c) return VLA_ERR_ARG;
  for (int r = 0; r < AR_MAX_RANKS; ++r)
    c->p.region[r] = r < world ? static_cast<unsigned char*>(regions[r]) : nullptr;
  c->p.rank = rank;
  c->p.world = world;
  c->p.max_bytes = max_bytes;
  c->num_blocks = num_blocks;
  hipError_t e = hipGetDevice(&c->device);
  int clock_khz = 0;
  if (e == hipSuccess)
    e = hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, c->device);
  if (e != hipSuccess) {
    delete c;
    VLA_HIP_CHECK(e);
  }
  // VLA_XGMI_WAIT_SECONDS: another bound for one barrier wait (tests of the timeout path)
  double seconds = AR_WAIT_SECONDS;
  if (const char* env = getenv("VLA_XGMI_WAIT_SECONDS")) {
    const double v = atof(env);
    if (v > 0) seconds = v;
  }
  if (clock_khz <= 0) clock_khz = 100000;   // the 100 MHz constant clock of the CDNA parts
  c->p.wait_ticks = (uint64_t)(seconds * 1000.0 * (double)clock_khz);
  *ctx = c;
  return VLA_OK;
}

extern "C" int vla_ar_destroy(void* ctx) {
  delete static_cast<ArContext*>(ctx);
  return VLA_OK;
}

extern "C" int vla_ar_all_reduce(void* ctx, void* out, const void* in, int64_t numel, int dtype,
                                 void* stream) {
  (void)hipGetLastError();
  ArContext* c = static_cast<ArContext*>(ctx);
  if (!c || !out || !in || numel < 0) return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16 && dtype != VLA_F32) return VLA_ERR_DTYPE;
  const int64_t bytes = numel * (dtype == VLA_F32 ? 4 : 2);
  if (bytes % 16 || bytes > c->p.max_bytes || ((uintptr_t)out % 16) || ((uintptr_t)in % 16))
    return VLA_ERR_ARG;
  const int64_t nvec = bytes / 16;
  const int64_t per = (nvec + c->p.world - 1) / c->p.world;
  dim3 grid(c->num_blocks), block(AR_THREADS);
  hipStream_t s = (hipStream_t)stream;
  // an empty message still runs the barriers: every rank must pass the same sequence
  if (dtype == VLA_F16)
    hipLaunchKernelGGL((xgmi_two_shot_kernel<F16>), grid, block, 0, s, c->p, (u32x4*)out,
                       (const u32x4*)in, nvec, per, 0, (int64_t)0);
  else if (dtype == VLA_BF16)
    hipLaunchKernelGGL((xgmi_two_shot_kernel<BF16>), grid, block, 0, s, c->p, (u32x4*)out,
                       (const u32x4*)in, nvec, per, 0, (int64_t)0);
  else
    hipLaunchKernelGGL((xgmi_two_shot_kernel<F32>), grid, block, 0, s, c->p, (u32x4*)out,
                       (const u32x4*)in, nvec, per, 0, (int64_t)0);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_ar_all_reduce_slabs(void* ctx, void* out, const float* slabs, int num_slabs,
                                       int64_t slab_stride, int64_t numel, int dtype,
                                       void* stream) {
  (void)hipGetLastError();
  ArContext* c = static_cast<ArContext*>(ctx);
  if (!c || !out || !slabs || numel < 0 || num_slabs < 1 || slab_stride < numel) return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16) return VLA_ERR_DTYPE;
  const int64_t bytes = numel * 2;
  if (bytes % 16 || bytes > c->p.max_bytes || ((uintptr_t)out % 16) || ((uintptr_t)slabs % 16) ||
      slab_stride % 4)
    return VLA_ERR_ARG;
  const int64_t nvec = bytes / 16;
  const int64_t per = (nvec + c->p.world - 1) / c->p.world;
  dim3 grid(c->num_blocks), block(AR_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VLA_F16)
    hipLaunchKernelGGL((xgmi_two_shot_kernel<F16, true>), grid, block, 0, s, c->p, (u32x4*)out,
                       (const u32x4*)slabs, nvec, per, num_slabs, slab_stride);
  else
    hipLaunchKernelGGL((xgmi_two_shot_kernel<BF16, true>), grid, block, 0, s, c->p, (u32x4*)out,
                       (const u32x4*)slabs, nvec, per, num_slabs, slab_stride);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_ar_all_reduce_add_rms_norm(void* ctx, void* out, const void* in, int num_slabs,
                                              int64_t slab_stride, void* residual,
                                              const void* weight, float epsilon, int rows,
                                              int hidden, int dtype, void* stream) {
  (void)hipGetLastError();
  ArContext* c = static_cast<ArContext*>(ctx);
  if (!c || !out || !in || !residual || !weight || rows < 0 || hidden <= 0 || num_slabs < 0)
    return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16) return VLA_ERR_DTYPE;
  const int64_t numel = (int64_t)rows * hidden;
  if (hidden % 8 || (hidden / 8) % c->p.world || hidden / 8 > 2 * AR_THREADS ||
      numel * 2 > c->p.max_bytes)
    return VLA_ERR_ARG;
  if (((uintptr_t)out | (uintptr_t)in | (uintptr_t)residual | (uintptr_t)weight) % 16)
    return VLA_ERR_ARG;
  if (num_slabs > 0 && (slab_stride < numel || slab_stride % 4)) return VLA_ERR_ARG;
  const int rv = hidden / 8;
  dim3 grid(c->num_blocks), block(AR_THREADS);
  hipStream_t s = (hipStream_t)stream;
#define VLA_AR_NORM(S, SL)                                                                      \
  do {                                                                                          \
    if (c->p.world == 2)                                                                        \
      hipLaunchKernelGGL((xgmi_two_shot_norm_kernel<S, SL, true>), grid, block, 0, s, c->p,     \
                         (u32x4*)out, (const u32x4*)in, (u32x4*)residual, (const u32x4*)weight, \
                         epsilon, rows, rv, num_slabs, slab_stride);                            \
    else                                                                                        \
      hipLaunchKernelGGL((xgmi_two_shot_norm_kernel<S, SL, false>), grid, block, 0, s, c->p,    \
                         (u32x4*)out, (const u32x4*)in, (u32x4*)residual, (const u32x4*)weight, \
                         epsilon, rows, rv, num_slabs, slab_stride);                            \
  } while (0)
  if (dtype == VLA_F16) {
    if (num_slabs > 0) VLA_AR_NORM(F16, true); else VLA_AR_NORM(F16, false);
  } else {
    if (num_slabs > 0) VLA_AR_NORM(BF16, true); else VLA_AR_NORM(BF16, false);
  }
#undef VLA_AR_NORM
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

// Asynchronous form of vla_ar_error_count: enqueues a 4-byte copy of the counter to `host_out`
// (pinned host memory) on `stream`; the value is there after the caller's next synchronisation.
extern "C" int vla_ar_error_count_async(void* ctx, int* host_out, void* stream) {
  ArContext* c = static_cast<ArContext*>(ctx);
  if (!c || !host_out) return VLA_ERR_ARG;
  VLA_HIP_CHECK(hipMemcpyAsync(host_out, c->p.region[c->p.rank] + AR_ERR_OFF, 4,
                               hipMemcpyDeviceToHost, (hipStream_t)stream));
  return VLA_OK;
}

// Synchronous: number of barrier waits that timed out on this rank since the region was made.
extern "C" int vla_ar_error_count(void* ctx, int* out) {
  ArContext* c = static_cast<ArContext*>(ctx);
  if (!c || !out) return VLA_ERR_ARG;
  uint32_t v = 0;
  VLA_HIP_CHECK(hipMemcpy(&v, c->p.region[c->p.rank] + AR_ERR_OFF, 4, hipMemcpyDeviceToHost));
  *out = (int)v;
  return VLA_OK;
}
