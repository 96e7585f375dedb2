// Kernel study (not part of the library): variants of the int4 dequant-GEMM main loop with
// ablation switches, timed standalone with HIP events.  Build: see tools/exp/Makefile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../vllm_legacy_amd/csrc/vla_common.h"

using namespace vla;

constexpr int TM = 256, TK = 64;
constexpr int X_TILE_BYTES = TM * TK * 2;

__device__ __forceinline__ f16x2 as_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ uint32_t as_u32(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t m, uint32_t o) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(m), "v"(o));
  return r;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
enum { ABL_NODEQ = 1, ABL_NOX = 2, ABL_NOW = 4, ABL_NOMFMA = 8, ABL_NOLDSREAD = 16 };

// V: 0 = pointer loads (as the library), 1 = buffer loads with scalar offsets + and_or asm
template <int ZOFF, int WN, int ABL, int V>
__global__ __launch_bounds__(128 * WN) void q4_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, _Float16* __restrict__ outp,
    int m, int n, int k, int group_size) {
  constexpr int THREADS = 128 * WN;
  constexpr int XCH = (TM * 8) / THREADS;
  constexpr int XROWS = THREADS / 8;
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * X_TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = blockIdx.x * (32 * WN), m0 = blockIdx.y * TM;
  const int k_begin = 0, k_end = k;
  const int nsteps = (k_end - k_begin) / TK;

  const int srow = tid >> 3, schunk = tid & 7;
  const int ncol = n0 + 32 * wn + c;
  const bool nok = ncol < n;
  const int ncl = nok ? ncol : 0;
  const int zshift = 4 * (ncl & 7);
  const int sshift = 16 * (ncl & 1);

  // buffer resources
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, m * k * 2, 0x00020000);
  __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)qw, 0, (k / 8) * n * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)qz, 0, (k / group_size) * (n / 8) * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)scales, 0, (k / group_size) * n * 2, 0x00020000);
  int xvoff[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) xvoff[i] = (min(m0 + srow + XROWS * i, m - 1) * k + 8 * schunk) * 2;
  const int wvoff = (h * n + ncl) * 4, zvoff = (ncl >> 3) * 4, svoff = (ncl >> 1) * 4;

  const _Float16* xsrc[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i)
    xsrc[i] = x + (int64_t)min(m0 + srow + XROWS * i, m - 1) * k + 8 * schunk;
  const uint32_t* wsrc = qw + (int64_t)h * n + ncl;
  const uint32_t* zsrc = qz + (ncl >> 3);
  const uint32_t* ssrc = reinterpret_cast<const uint32_t*>(scales) + (ncl >> 1);

  u32x4 xr[XCH];
  auto load_x = [&](int k0) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      if constexpr (V == 0) xr[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + k0);
      else xr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xvoff[i], k0 * 2, 0));
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int r = srow + XROWS * i;
      *reinterpret_cast<u32x4*>(xs + buf * X_TILE_BYTES + r * 128 + ((schunk ^ ((r >> 1) & 7)) << 4)) = xr[i];
    }
  };
  struct WStep { uint32_t w[4]; uint32_t z; uint32_t s; };
  auto load_w = [&](int step, WStep& d) {
    step = min(step, nsteps - 1);
    const int k0 = k_begin + step * TK;
    const int grp = k0 / group_size;
    if constexpr (V == 0) {
      const uint32_t* p = wsrc + (int64_t)(k0 >> 3) * n;
#pragma unroll
      for (int s = 0; s < 4; ++s) d.w[s] = p[(int64_t)(2 * s) * n];
      d.z = zsrc[(int64_t)grp * (n >> 3)];
      d.s = ssrc[(int64_t)grp * (n >> 1)];
    } else {
      const int so = (k0 >> 3) * n * 4;
#pragma unroll
      for (int s = 0; s < 4; ++s) d.w[s] = __builtin_amdgcn_raw_buffer_load_b32(rw, wvoff, so + 2 * s * n * 4, 0);
      d.z = __builtin_amdgcn_raw_buffer_load_b32(rz, zvoff, grp * (n >> 3) * 4, 0);
      d.s = __builtin_amdgcn_raw_buffer_load_b32(rs, svoff, grp * (n >> 1) * 4, 0);
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  uint32_t mlo = 0x000f000fu, mhi = 0x00f000f0u, magic = 0x64006400u;
  if constexpr (V == 1) { asm volatile("" : "+v"(mlo), "+v"(mhi), "+v"(magic)); }

  auto k_step = [&](int t, const WStep& cur, WStep& nxt) {
    const bool has_next = t + 1 < nsteps;
    const int buf = t & 1;
    if constexpr (!(ABL & ABL_NOX)) { if (has_next) load_x(k_begin + (t + 1) * TK); }
    if constexpr (!(ABL & ABL_NOW)) load_w(t + 2, nxt);
    __builtin_amdgcn_sched_barrier(0);

    f16x2 zc1, zc2, sc2;
    if constexpr (V == 0) {
      const int z = (int)((cur.z >> zshift) & 15u) + ZOFF;
      const _Float16 z1 = (_Float16)(float)(-(1024 + z)), z2 = (_Float16)(float)(-(64 + z));
      zc1 = f16x2{z1, z1}; zc2 = f16x2{z2, z2};
    } else {
      const uint32_t z = ((cur.z >> zshift) & 15u) + ZOFF;       // 0..16
      zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);                // -(1024 + z): ulp 1 there
      zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);                // -(64 + z): ulp 1/16 there
    }
    sc2 = as_h2(__builtin_amdgcn_perm(cur.s, cur.s, sshift ? 0x03020302u : 0x01000100u));

    const unsigned char* xb = xs + buf * X_TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint32_t w = cur.w[s], w8 = w >> 8;
      u32x4 bf;
      if constexpr (ABL & ABL_NODEQ) {
        bf = u32x4{w, w8, w, w8};
      } else {
        f16x2 d0, d1, d2, d3;
        if constexpr (V == 0) {
          d0 = as_h2((w & 0x000f000fu) | 0x64006400u) + zc1;
          d1 = __builtin_elementwise_fma(as_h2((w & 0x00f000f0u) | 0x64006400u), sixteenth, zc2);
          d2 = as_h2((w8 & 0x000f000fu) | 0x64006400u) + zc1;
          d3 = __builtin_elementwise_fma(as_h2((w8 & 0x00f000f0u) | 0x64006400u), sixteenth, zc2);
        } else {
          d0 = as_h2(and_or(w, mlo, magic)) + zc1;
          d1 = __builtin_elementwise_fma(as_h2(and_or(w, mhi, magic)), sixteenth, zc2);
          d2 = as_h2(and_or(w8, mlo, magic)) + zc1;
          d3 = __builtin_elementwise_fma(as_h2(and_or(w8, mhi, magic)), sixteenth, zc2);
        }
        bf[0] = as_u32(d0 * sc2); bf[1] = as_u32(d1 * sc2);
        bf[2] = as_u32(d2 * sc2); bf[3] = as_u32(d3 * sc2);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int row = 128 * wm + 32 * a + c;
        u32x4 af;
        if constexpr (ABL & ABL_NOLDSREAD) af = u32x4{(uint32_t)row, w, w8, (uint32_t)s};
        else af = *reinterpret_cast<const u32x4*>(xb + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4));
        if constexpr (ABL & ABL_NOMFMA) {
          acc[a][0] += __builtin_bit_cast(float, af[0] ^ bf[0]);
          acc[a][1] += __builtin_bit_cast(float, af[1] ^ bf[1]);
          acc[a][2] += __builtin_bit_cast(float, af[2] ^ bf[2]);
          acc[a][3] += __builtin_bit_cast(float, af[3] ^ bf[3]);
        } else {
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af),
                                                          __builtin_bit_cast(f16x8, bf), acc[a], 0, 0, 0);
        }
      }
    }
    if constexpr (!(ABL & ABL_NOX)) { if (has_next) store_x(buf ^ 1); }
    __syncthreads();
  };

  WStep w0, w1, w2;
  load_x(k_begin);
  load_w(0, w0);
  load_w(1, w1);
  w2 = w1;
  store_x(0);
  if constexpr (ABL & ABL_NOX) store_x(1);
  __syncthreads();

  int t = 0;
  for (; t + 3 <= nsteps; t += 3) {
    k_step(t, w0, w2);
    k_step(t + 1, w1, w0);
    k_step(t + 2, w2, w1);
  }
  if (t < nsteps) k_step(t, w0, w2);
  if (t + 1 < nsteps) k_step(t + 1, w1, w0);

  if (!nok) return;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + 128 * wm + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row >= m) continue;
      outp[(int64_t)row * n + ncol] = (_Float16)acc[a][i];
    }
}



// Variant 2: x two K steps ahead in two register sets, weights three steps ahead in a four-name
// ring, loop unrolled by four.
template <int ZOFF, int WN, int ABL>
__global__ __launch_bounds__(128 * WN) void q4_kernel2(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, _Float16* __restrict__ outp,
    int m, int n, int k, int group_size) {
  constexpr int THREADS = 128 * WN;
  constexpr int XCH = (TM * 8) / THREADS;
  constexpr int XROWS = THREADS / 8;
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * X_TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = blockIdx.x * (32 * WN), m0 = blockIdx.y * TM;
  const int k_begin = 0, k_end = k;
  const int nsteps = (k_end - k_begin) / TK;

  const int srow = tid >> 3, schunk = tid & 7;
  const int ncol = n0 + 32 * wn + c;
  const bool nok = ncol < n;
  const int ncl = nok ? ncol : 0;
  const int zshift = 4 * (ncl & 7);
  const int sshift = 16 * (ncl & 1);

  const _Float16* xsrc[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i)
    xsrc[i] = x + (int64_t)min(m0 + srow + XROWS * i, m - 1) * k + 8 * schunk;
  const uint32_t* wsrc = qw + (int64_t)h * n + ncl;
  const uint32_t* zsrc = qz + (ncl >> 3);
  const uint32_t* ssrc = reinterpret_cast<const uint32_t*>(scales) + (ncl >> 1);

  struct XSet { u32x4 v[XCH]; };
  auto load_x = [&](int step, XSet& d) {
    const int k0 = k_begin + min(step, nsteps - 1) * TK;
#pragma unroll
    for (int i = 0; i < XCH; ++i) d.v[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + k0);
  };
  auto store_x = [&](int buf, const XSet& d) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int r = srow + XROWS * i;
      *reinterpret_cast<u32x4*>(xs + buf * X_TILE_BYTES + r * 128 + ((schunk ^ ((r >> 1) & 7)) << 4)) = d.v[i];
    }
  };
  struct WStep { uint32_t w[4]; uint32_t z; uint32_t s; };
  auto load_w = [&](int step, WStep& d) {
    step = min(step, nsteps - 1);
    const int k0 = k_begin + step * TK;
    const int grp = k0 / group_size;
    const uint32_t* p = wsrc + (int64_t)(k0 >> 3) * n;
#pragma unroll
    for (int s = 0; s < 4; ++s) d.w[s] = p[(int64_t)(2 * s) * n];
    d.z = zsrc[(int64_t)grp * (n >> 3)];
    d.s = ssrc[(int64_t)grp * (n >> 1)];
  };

  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};

  // step t: x(t+2) -> xnew (the set whose tile was parked at the end of step t-1), w(t+3) -> wnew,
  // compute from wcur and LDS buffer t & 1, park xpark = x(t+1) into the other buffer.
  auto k_step = [&](int t, const WStep& wcur, WStep& wnew, XSet& xnew, const XSet& xpark) {
    const int buf = t & 1;
    if constexpr (!(ABL & ABL_NOX)) load_x(t + 2, xnew);
    if constexpr (!(ABL & ABL_NOW)) load_w(t + 3, wnew);
    __builtin_amdgcn_sched_barrier(0);

    const uint32_t z = ((wcur.z >> zshift) & 15u) + ZOFF;
    const f16x2 zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);
    const f16x2 zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);
    const f16x2 sc2 = as_h2(__builtin_amdgcn_perm(wcur.s, wcur.s, sshift ? 0x03020302u : 0x01000100u));

    const unsigned char* xb = xs + buf * X_TILE_BYTES;
    auto read_a = [&](int s, u32x4 (&af)[4]) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int row = 128 * wm + 32 * a + c;
        af[a] = *reinterpret_cast<const u32x4*>(xb + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4));
      }
    };
    auto dequant = [&](int s) {
      const uint32_t w = wcur.w[s], w8 = w >> 8;
      u32x4 bf;
      if constexpr (ABL & ABL_NODEQ) {
        bf = u32x4{w, w8, w, w8};
      } else {
        const f16x2 d0 = as_h2((w & 0x000f000fu) | 0x64006400u) + zc1;
        const f16x2 d1 = __builtin_elementwise_fma(as_h2((w & 0x00f000f0u) | 0x64006400u), sixteenth, zc2);
        const f16x2 d2 = as_h2((w8 & 0x000f000fu) | 0x64006400u) + zc1;
        const f16x2 d3 = __builtin_elementwise_fma(as_h2((w8 & 0x00f000f0u) | 0x64006400u), sixteenth, zc2);
        bf[0] = as_u32(d0 * sc2); bf[1] = as_u32(d1 * sc2);
        bf[2] = as_u32(d2 * sc2); bf[3] = as_u32(d3 * sc2);
      }
      return bf;
    };
    auto mma = [&](const u32x4 (&af)[4], const u32x4& bf) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[a]),
                                                        __builtin_bit_cast(f16x8, bf), acc[a], 0, 0, 0);
    };
#define SB __builtin_amdgcn_sched_barrier(0)
    u32x4 fa[4], fb[4];
    read_a(0, fa);
    u32x4 b0 = dequant(0);
    SB;
    read_a(1, fb);
    SB;
    mma(fa, b0);
    u32x4 b1 = dequant(1);
    SB;
    read_a(2, fa);
    SB;
    mma(fb, b1);
    b0 = dequant(2);
    SB;
    read_a(3, fb);
    SB;
    mma(fa, b0);
    b1 = dequant(3);
    SB;
    mma(fb, b1);
#undef SB
    if constexpr (!(ABL & ABL_NOX)) { if (t + 1 < nsteps) store_x(buf ^ 1, xpark); }
    __syncthreads();
  };

  WStep w0, w1, w2, w3;
  XSet xa, xb_;
  load_x(0, xa);
  load_x(1, xb_);
  load_w(0, w0);
  load_w(1, w1);
  load_w(2, w2);
  w3 = w2;
  store_x(0, xa);
  if constexpr (ABL & ABL_NOX) store_x(1, xb_);
  __syncthreads();

  int t = 0;
  for (; t + 4 <= nsteps; t += 4) {
    k_step(t, w0, w3, xa, xb_);
    k_step(t + 1, w1, w0, xb_, xa);
    k_step(t + 2, w2, w1, xa, xb_);
    k_step(t + 3, w3, w2, xb_, xa);
  }
  if (t < nsteps) k_step(t, w0, w3, xa, xb_);
  if (t + 1 < nsteps) k_step(t + 1, w1, w0, xb_, xa);
  if (t + 2 < nsteps) k_step(t + 2, w2, w1, xa, xb_);

  if (!nok) return;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + 128 * wm + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row >= m) continue;
      outp[(int64_t)row * n + ncol] = (_Float16)acc[a][i];
    }
}

// Variant 4 = variant 2 with optimisation switches OPT (1: uniform-base + 32-bit lane offset
// addressing, 2: mask constants held in VGPRs so that and+or fuse, 4: MFMA/VALU interleave hints): x two K steps ahead in two register sets, weights three steps ahead in a four-name
// ring, loop unrolled by four.
template <int ZOFF, int WN, int ABL, int OPT>
__global__ __launch_bounds__(128 * WN) void q4_kernel4(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, _Float16* __restrict__ outp,
    int m, int n, int k, int group_size) {
  constexpr int THREADS = 128 * WN;
  constexpr int XCH = (TM * 8) / THREADS;
  constexpr int XROWS = THREADS / 8;
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * X_TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = blockIdx.x * (32 * WN), m0 = blockIdx.y * TM;
  const int k_begin = 0, k_end = k;
  const int nsteps = (k_end - k_begin) / TK;

  const int srow = tid >> 3, schunk = tid & 7;
  const int ncol = n0 + 32 * wn + c;
  const bool nok = ncol < n;
  const int ncl = nok ? ncol : 0;
  const int zshift = 4 * (ncl & 7);
  const int sshift = 16 * (ncl & 1);

  const _Float16* xsrc[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i)
    xsrc[i] = x + (int64_t)min(m0 + srow + XROWS * i, m - 1) * k + 8 * schunk;
  const uint32_t* wsrc = qw + (int64_t)h * n + ncl;
  const uint32_t* zsrc = qz + (ncl >> 3);
  const uint32_t* ssrc = reinterpret_cast<const uint32_t*>(scales) + (ncl >> 1);

  struct XSet { u32x4 v[XCH]; };
  uint32_t xoff[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) xoff[i] = (uint32_t)((min(m0 + srow + XROWS * i, m - 1) * k + 8 * schunk) * 2);
  const uint32_t woff = (uint32_t)((h * n + ncl) * 4), zoff = (uint32_t)((ncl >> 3) * 4), soff = (uint32_t)((ncl >> 1) * 4);
  auto load_x = [&](int step, XSet& d) {
    const int k0 = k_begin + min(step, nsteps - 1) * TK;
    if constexpr (OPT & 1) {
      const char* base = reinterpret_cast<const char*>(x + k0);
#pragma unroll
      for (int i = 0; i < XCH; ++i) d.v[i] = *reinterpret_cast<const u32x4*>(base + xoff[i]);
    } else {
#pragma unroll
      for (int i = 0; i < XCH; ++i) d.v[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + k0);
    }
  };
  auto store_x = [&](int buf, const XSet& d) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int r = srow + XROWS * i;
      *reinterpret_cast<u32x4*>(xs + buf * X_TILE_BYTES + r * 128 + ((schunk ^ ((r >> 1) & 7)) << 4)) = d.v[i];
    }
  };
  struct WStep { uint32_t w[4]; uint32_t z; uint32_t s; };
  auto load_w = [&](int step, WStep& d) {
    step = min(step, nsteps - 1);
    const int k0 = k_begin + step * TK;
    const int grp = k0 / group_size;
    if constexpr (OPT & 1) {
      const char* pb = reinterpret_cast<const char*>(qw + (int64_t)(k0 >> 3) * n);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        d.w[s] = *reinterpret_cast<const uint32_t*>(pb + (int64_t)(2 * s) * n * 4 + woff);
      d.z = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(qz + (int64_t)grp * (n >> 3)) + zoff);
      d.s = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(scales + (int64_t)grp * n) + soff);
    } else {
      const uint32_t* p = wsrc + (int64_t)(k0 >> 3) * n;
#pragma unroll
      for (int s = 0; s < 4; ++s) d.w[s] = p[(int64_t)(2 * s) * n];
      d.z = zsrc[(int64_t)grp * (n >> 3)];
      d.s = ssrc[(int64_t)grp * (n >> 1)];
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  uint32_t mlo = 0x000f000fu, mhi = 0x00f000f0u, magic = 0x64006400u;
  if constexpr (OPT & 2) { asm volatile("" : "+v"(mlo), "+v"(mhi), "+v"(magic)); }

  // step t: x(t+2) -> xnew (the set whose tile was parked at the end of step t-1), w(t+3) -> wnew,
  // compute from wcur and LDS buffer t & 1, park xpark = x(t+1) into the other buffer.
  auto k_step = [&](int t, const WStep& wcur, WStep& wnew, XSet& xnew, const XSet& xpark) {
    const int buf = t & 1;
    if constexpr (!(ABL & ABL_NOX)) load_x(t + 2, xnew);
    if constexpr (!(ABL & ABL_NOW)) load_w(t + 3, wnew);
    __builtin_amdgcn_sched_barrier(0);

    const uint32_t z = ((wcur.z >> zshift) & 15u) + ZOFF;
    const f16x2 zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);
    const f16x2 zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);
    const f16x2 sc2 = as_h2(__builtin_amdgcn_perm(wcur.s, wcur.s, sshift ? 0x03020302u : 0x01000100u));

    const unsigned char* xb = xs + buf * X_TILE_BYTES;
    auto read_a = [&](int s, u32x4 (&af)[4]) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int row = 128 * wm + 32 * a + c;
        af[a] = *reinterpret_cast<const u32x4*>(xb + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4));
      }
    };
    auto dequant = [&](int s) {
      const uint32_t w = wcur.w[s], w8 = w >> 8;
      u32x4 bf;
      if constexpr (ABL & ABL_NODEQ) {
        bf = u32x4{w, w8, w, w8};
      } else {
        const f16x2 d0 = as_h2((w & mlo) | magic) + zc1;
        const f16x2 d1 = __builtin_elementwise_fma(as_h2((w & mhi) | magic), sixteenth, zc2);
        const f16x2 d2 = as_h2((w8 & mlo) | magic) + zc1;
        const f16x2 d3 = __builtin_elementwise_fma(as_h2((w8 & mhi) | magic), sixteenth, zc2);
        bf[0] = as_u32(d0 * sc2); bf[1] = as_u32(d1 * sc2);
        bf[2] = as_u32(d2 * sc2); bf[3] = as_u32(d3 * sc2);
      }
      return bf;
    };
    auto mma = [&](const u32x4 (&af)[4], const u32x4& bf) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[a]),
                                                        __builtin_bit_cast(f16x8, bf), acc[a], 0, 0, 0);
    };
    u32x4 fa[4], fb[4];
    if constexpr (OPT & 4) {
      read_a(0, fa);
      u32x4 b0 = dequant(0);
      read_a(1, fb);
      u32x4 b1 = dequant(1);
      mma(fa, b0);
      read_a(2, fa);
      b0 = dequant(2);
      mma(fb, b1);
      read_a(3, fb);
      b1 = dequant(3);
      mma(fa, b0);
      mma(fb, b1);
      // issue order: 8 fragment reads and the first two dequantisations, then each MFMA followed
      // by a few vector ops and, while reads remain, one fragment read
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 34, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
    } else {
#define SB __builtin_amdgcn_sched_barrier(0)
      read_a(0, fa);
      u32x4 b0 = dequant(0);
      SB;
      read_a(1, fb);
      SB;
      mma(fa, b0);
      u32x4 b1 = dequant(1);
      SB;
      read_a(2, fa);
      SB;
      mma(fb, b1);
      b0 = dequant(2);
      SB;
      read_a(3, fb);
      SB;
      mma(fa, b0);
      b1 = dequant(3);
      SB;
      mma(fb, b1);
#undef SB
    }
    if constexpr (!(ABL & ABL_NOX)) { if (t + 1 < nsteps) store_x(buf ^ 1, xpark); }
    __syncthreads();
  };

  WStep w0, w1, w2, w3;
  XSet xa, xb_;
  load_x(0, xa);
  load_x(1, xb_);
  load_w(0, w0);
  load_w(1, w1);
  load_w(2, w2);
  w3 = w2;
  store_x(0, xa);
  if constexpr (ABL & ABL_NOX) store_x(1, xb_);
  __syncthreads();

  int t = 0;
  for (; t + 4 <= nsteps; t += 4) {
    k_step(t, w0, w3, xa, xb_);
    k_step(t + 1, w1, w0, xb_, xa);
    k_step(t + 2, w2, w1, xa, xb_);
    k_step(t + 3, w3, w2, xb_, xa);
  }
  if (t < nsteps) k_step(t, w0, w3, xa, xb_);
  if (t + 1 < nsteps) k_step(t + 1, w1, w0, xb_, xa);
  if (t + 2 < nsteps) k_step(t + 2, w2, w1, xa, xb_);

  if (!nok) return;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + 128 * wm + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row >= m) continue;
      outp[(int64_t)row * n + ncol] = (_Float16)acc[a][i];
    }
}

// Variant 5: 128 (M) x 256 (N) workgroup tile, 8 waves along N (no duplicated dequantisation,
// half the x traffic per flop); otherwise variant 4 = variant 2 with optimisation switches OPT (1: uniform-base + 32-bit lane offset
// addressing, 2: mask constants held in VGPRs so that and+or fuse, 4: MFMA/VALU interleave hints): x two K steps ahead in two register sets, weights three steps ahead in a four-name
// ring, loop unrolled by four.
template <int ZOFF, int WN, int ABL, int OPT>
__global__ __launch_bounds__(128 * WN) void q4_kernel5(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, _Float16* __restrict__ outp,
    int m, int n, int k, int group_size) {
  constexpr int THREADS = 128 * WN;
  constexpr int TM5 = 128;
  constexpr int XT = TM5 * TK * 2;
  constexpr int XCH = (TM5 * 8) / THREADS;
  constexpr int XROWS = THREADS / 8;
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * XT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int wm = 0, wn = wave;
  const int n0 = blockIdx.x * (64 * WN), m0 = blockIdx.y * TM5;
  const int k_begin = 0, k_end = k;
  const int nsteps = (k_end - k_begin) / TK;

  const int srow = tid >> 3, schunk = tid & 7;
  const int ncol = n0 + 32 * wn + c;
  const bool nok = ncol < n;
  const int ncl = nok ? ncol : 0;
  const int zshift = 4 * (ncl & 7);
  const int sshift = 16 * (ncl & 1);

  const _Float16* xsrc[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i)
    xsrc[i] = x + (int64_t)min(m0 + srow + XROWS * i, m - 1) * k + 8 * schunk;
  const uint32_t* wsrc = qw + (int64_t)h * n + ncl;
  const uint32_t* zsrc = qz + (ncl >> 3);
  const uint32_t* ssrc = reinterpret_cast<const uint32_t*>(scales) + (ncl >> 1);

  struct XSet { u32x4 v[XCH]; };
  uint32_t xoff[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) xoff[i] = (uint32_t)((min(m0 + srow + XROWS * i, m - 1) * k + 8 * schunk) * 2);
  const uint32_t woff = (uint32_t)((h * n + ncl) * 4), zoff = (uint32_t)((ncl >> 3) * 4), soff = (uint32_t)((ncl >> 1) * 4);
  auto load_x = [&](int step, XSet& d) {
    const int k0 = k_begin + min(step, nsteps - 1) * TK;
    if constexpr (OPT & 1) {
      const char* base = reinterpret_cast<const char*>(x + k0);
#pragma unroll
      for (int i = 0; i < XCH; ++i) d.v[i] = *reinterpret_cast<const u32x4*>(base + xoff[i]);
    } else {
#pragma unroll
      for (int i = 0; i < XCH; ++i) d.v[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + k0);
    }
  };
  auto store_x = [&](int buf, const XSet& d) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int r = srow + XROWS * i;
      *reinterpret_cast<u32x4*>(xs + buf * XT + r * 128 + ((schunk ^ ((r >> 1) & 7)) << 4)) = d.v[i];
    }
  };
  struct WStep { uint32_t w[4]; uint32_t z; uint32_t s; };
  auto load_w = [&](int step, WStep& d) {
    step = min(step, nsteps - 1);
    const int k0 = k_begin + step * TK;
    const int grp = k0 / group_size;
    if constexpr (OPT & 16) {
      // timing only: one 16-byte load per lane, as a [K/64][2][N][4]-word layout would allow
      const u32x4 wv = *reinterpret_cast<const u32x4*>(qw + ((int64_t)(2 * (k0 / TK) + h) * n + ncl) * 4);
      d.w[0] = wv[0]; d.w[1] = wv[1]; d.w[2] = wv[2]; d.w[3] = wv[3];
      if constexpr (OPT & 32) { d.z = 0x88888888u; d.s = 0x21002100u; }
      else {
        d.z = zsrc[(int64_t)grp * (n >> 3)];
        d.s = ssrc[(int64_t)grp * (n >> 1)];
      }
    } else if constexpr (OPT & 1) {
      const char* pb = reinterpret_cast<const char*>(qw + (int64_t)(k0 >> 3) * n);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        d.w[s] = *reinterpret_cast<const uint32_t*>(pb + (int64_t)(2 * s) * n * 4 + woff);
      d.z = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(qz + (int64_t)grp * (n >> 3)) + zoff);
      d.s = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(scales + (int64_t)grp * n) + soff);
    } else {
      const uint32_t* p = wsrc + (int64_t)(k0 >> 3) * n;
#pragma unroll
      for (int s = 0; s < 4; ++s) d.w[s] = p[(int64_t)(2 * s) * n];
      d.z = zsrc[(int64_t)grp * (n >> 3)];
      d.s = ssrc[(int64_t)grp * (n >> 1)];
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  uint32_t mlo = 0x000f000fu, mhi = 0x00f000f0u, magic = 0x64006400u;
  if constexpr (OPT & 2) { asm volatile("" : "+v"(mlo), "+v"(mhi), "+v"(magic)); }

  // step t: x(t+2) -> xnew (the set whose tile was parked at the end of step t-1), w(t+3) -> wnew,
  // compute from wcur and LDS buffer t & 1, park xpark = x(t+1) into the other buffer.
  auto k_step = [&](int t, const WStep& wcur, WStep& wnew, XSet& xnew, const XSet& xpark) {
    const int buf = t & 1;
    if constexpr (!(ABL & ABL_NOX)) load_x(t + 2, xnew);
    if constexpr (!(ABL & ABL_NOW)) load_w(t + 3, wnew);
    __builtin_amdgcn_sched_barrier(0);

    const uint32_t z = ((wcur.z >> zshift) & 15u) + ZOFF;
    const f16x2 zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);
    const f16x2 zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);
    const f16x2 sc2 = as_h2(__builtin_amdgcn_perm(wcur.s, wcur.s, sshift ? 0x03020302u : 0x01000100u));

    const unsigned char* xb = xs + buf * XT;
    auto read_a = [&](int s, u32x4 (&af)[4]) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int row = 128 * wm + 32 * a + c;
        af[a] = *reinterpret_cast<const u32x4*>(xb + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4));
      }
    };
    auto dequant = [&](int s) {
      const uint32_t w = wcur.w[s], w8 = w >> 8;
      u32x4 bf;
      if constexpr (ABL & ABL_NODEQ) {
        bf = u32x4{w, w8, w, w8};
      } else {
        const f16x2 d0 = as_h2((w & mlo) | magic) + zc1;
        const f16x2 d1 = __builtin_elementwise_fma(as_h2((w & mhi) | magic), sixteenth, zc2);
        const f16x2 d2 = as_h2((w8 & mlo) | magic) + zc1;
        const f16x2 d3 = __builtin_elementwise_fma(as_h2((w8 & mhi) | magic), sixteenth, zc2);
        bf[0] = as_u32(d0 * sc2); bf[1] = as_u32(d1 * sc2);
        bf[2] = as_u32(d2 * sc2); bf[3] = as_u32(d3 * sc2);
      }
      return bf;
    };
    auto mma = [&](const u32x4 (&af)[4], const u32x4& bf) {
      if constexpr (OPT & 8) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int a = 0; a < 4; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[a]),
                                                        __builtin_bit_cast(f16x8, bf), acc[a], 0, 0, 0);
      if constexpr (OPT & 8) __builtin_amdgcn_s_setprio(0);
    };
    u32x4 fa[4], fb[4];
    if constexpr (OPT & 4) {
      read_a(0, fa);
      u32x4 b0 = dequant(0);
      read_a(1, fb);
      u32x4 b1 = dequant(1);
      mma(fa, b0);
      read_a(2, fa);
      b0 = dequant(2);
      mma(fb, b1);
      read_a(3, fb);
      b1 = dequant(3);
      mma(fa, b0);
      mma(fb, b1);
      // issue order: 8 fragment reads and the first two dequantisations, then each MFMA followed
      // by a few vector ops and, while reads remain, one fragment read
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 34, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
    } else {
#define SB __builtin_amdgcn_sched_barrier(0)
      read_a(0, fa);
      u32x4 b0 = dequant(0);
      SB;
      read_a(1, fb);
      SB;
      mma(fa, b0);
      u32x4 b1 = dequant(1);
      SB;
      read_a(2, fa);
      SB;
      mma(fb, b1);
      b0 = dequant(2);
      SB;
      read_a(3, fb);
      SB;
      mma(fa, b0);
      b1 = dequant(3);
      SB;
      mma(fb, b1);
#undef SB
    }
    if constexpr (!(ABL & ABL_NOX)) { if (t + 1 < nsteps) store_x(buf ^ 1, xpark); }
    __syncthreads();
  };

  WStep w0, w1, w2, w3;
  XSet xa, xb_;
  load_x(0, xa);
  load_x(1, xb_);
  load_w(0, w0);
  load_w(1, w1);
  load_w(2, w2);
  w3 = w2;
  store_x(0, xa);
  if constexpr (ABL & ABL_NOX) store_x(1, xb_);
  __syncthreads();

  int t = 0;
  for (; t + 4 <= nsteps; t += 4) {
    k_step(t, w0, w3, xa, xb_);
    k_step(t + 1, w1, w0, xb_, xa);
    k_step(t + 2, w2, w1, xa, xb_);
    k_step(t + 3, w3, w2, xb_, xa);
  }
  if (t < nsteps) k_step(t, w0, w3, xa, xb_);
  if (t + 1 < nsteps) k_step(t + 1, w1, w0, xb_, xa);
  if (t + 2 < nsteps) k_step(t + 2, w2, w1, xa, xb_);

  if (!nok) return;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + 128 * wm + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row >= m) continue;
      outp[(int64_t)row * n + ncol] = (_Float16)acc[a][i];
    }
}

template <int ABL, int OPT>
static float run5(const char* name, int m, int n, int k, _Float16* x, std::vector<uint32_t*>& qw, uint32_t* qz,
                  _Float16* sc, _Float16* out, int iters) {
  dim3 grid((n + 255) / 256, (m + 127) / 128, 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((q4_kernel5<0, 4, ABL, OPT>), grid, dim3(512), 0, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((q4_kernel5<0, 4, ABL, OPT>), grid, dim3(512), 0, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const float us = ms * 1e3f / iters;
  printf("%-28s n=%5d k=%5d  %8.1f us  %7.1f TFLOP/s\n", name, n, k, us, 2.0 * m * n * k / (us * 1e-6) / 1e12);
  return us;
}
template <int ABL, int OPT>
static float run4(const char* name, int m, int n, int k, _Float16* x, std::vector<uint32_t*>& qw, uint32_t* qz,
                  _Float16* sc, _Float16* out, int iters) {
  dim3 grid((n + 127) / 128, (m + 255) / 256, 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((q4_kernel4<0, 4, ABL, OPT>), grid, dim3(512), 0, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((q4_kernel4<0, 4, ABL, OPT>), grid, dim3(512), 0, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const float us = ms * 1e3f / iters;
  printf("%-28s n=%5d k=%5d  %8.1f us  %7.1f TFLOP/s\n", name, n, k, us, 2.0 * m * n * k / (us * 1e-6) / 1e12);
  return us;
}
template <int ABL>
static float run2(const char* name, int m, int n, int k, _Float16* x, std::vector<uint32_t*>& qw, uint32_t* qz,
                  _Float16* sc, _Float16* out, int iters) {
  dim3 grid((n + 127) / 128, (m + 255) / 256, 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((q4_kernel2<0, 4, ABL>), grid, dim3(512), 0, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((q4_kernel2<0, 4, ABL>), grid, dim3(512), 0, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const float us = ms * 1e3f / iters;
  printf("%-28s n=%5d k=%5d  %8.1f us  %7.1f TFLOP/s\n", name, n, k, us, 2.0 * m * n * k / (us * 1e-6) / 1e12);
  return us;
}


template <int ABL, int V>
static float run(const char* name, int m, int n, int k, _Float16* x, std::vector<uint32_t*>& qw, uint32_t* qz,
                 _Float16* sc, _Float16* out, int iters) {
  dim3 grid((n + 127) / 128, (m + 255) / 256, 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((q4_kernel<0, 4, ABL, V>), grid, dim3(512), 0, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((q4_kernel<0, 4, ABL, V>), grid, dim3(512), 0, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const float us = ms * 1e3f / iters;
  printf("%-28s n=%5d k=%5d  %8.1f us  %7.1f TFLOP/s\n", name, n, k, us, 2.0 * m * n * k / (us * 1e-6) / 1e12);
  return us;
}

// Variant 3 (three LDS buffers, x parked early in the step, barrier mid-step, first A fragments
// of the next tile fetched before the step ends); derived from variant 2: x two K steps ahead in two register sets, weights three steps ahead in a four-name
// ring, loop unrolled by four.
template <int ZOFF, int WN, int ABL>
__global__ __launch_bounds__(128 * WN) void q4_kernel3(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, _Float16* __restrict__ outp,
    int m, int n, int k, int group_size) {
  constexpr int THREADS = 128 * WN;
  constexpr int XCH = (TM * 8) / THREADS;
  constexpr int XROWS = THREADS / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char xs[];   // 3 * X_TILE_BYTES

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = blockIdx.x * (32 * WN), m0 = blockIdx.y * TM;
  const int k_begin = 0, k_end = k;
  const int nsteps = (k_end - k_begin) / TK;

  const int srow = tid >> 3, schunk = tid & 7;
  const int ncol = n0 + 32 * wn + c;
  const bool nok = ncol < n;
  const int ncl = nok ? ncol : 0;
  const int zshift = 4 * (ncl & 7);
  const int sshift = 16 * (ncl & 1);

  const _Float16* xsrc[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i)
    xsrc[i] = x + (int64_t)min(m0 + srow + XROWS * i, m - 1) * k + 8 * schunk;
  const uint32_t* wsrc = qw + (int64_t)h * n + ncl;
  const uint32_t* zsrc = qz + (ncl >> 3);
  const uint32_t* ssrc = reinterpret_cast<const uint32_t*>(scales) + (ncl >> 1);

  struct XSet { u32x4 v[XCH]; };
  auto load_x = [&](int step, XSet& d) {
    const int k0 = k_begin + min(step, nsteps - 1) * TK;
#pragma unroll
    for (int i = 0; i < XCH; ++i) d.v[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + k0);
  };
  auto store_x = [&](int buf, const XSet& d) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int r = srow + XROWS * i;
      *reinterpret_cast<u32x4*>(xs + buf * X_TILE_BYTES + r * 128 + ((schunk ^ ((r >> 1) & 7)) << 4)) = d.v[i];
    }
  };
  struct WStep { uint32_t w[4]; uint32_t z; uint32_t s; };
  auto load_w = [&](int step, WStep& d) {
    step = min(step, nsteps - 1);
    const int k0 = k_begin + step * TK;
    const int grp = k0 / group_size;
    const uint32_t* p = wsrc + (int64_t)(k0 >> 3) * n;
#pragma unroll
    for (int s = 0; s < 4; ++s) d.w[s] = p[(int64_t)(2 * s) * n];
    d.z = zsrc[(int64_t)grp * (n >> 3)];
    d.s = ssrc[(int64_t)grp * (n >> 1)];
  };

  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};

  // step t: x(t+2) -> xnew (the set whose tile was parked at the end of step t-1), w(t+3) -> wnew,
  // compute from wcur and LDS buffer t & 1, park xpark = x(t+1) into the other buffer.
  u32x4 fa[4], fb[4];
  auto read_a = [&](int buf, int s, u32x4 (&af)[4]) {
    const unsigned char* xb = xs + buf * X_TILE_BYTES;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = 128 * wm + 32 * a + c;
      af[a] = *reinterpret_cast<const u32x4*>(xb + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4));
    }
  };
  auto mma = [&](const u32x4 (&af)[4], const u32x4& bf) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[a]),
                                                      __builtin_bit_cast(f16x8, bf), acc[a], 0, 0, 0);
  };
  int bcur = 0;   // LDS buffer of tile t
#define SB __builtin_amdgcn_sched_barrier(0)
  auto k_step = [&](int t, const WStep& wcur, WStep& wnew, XSet& xnew, const XSet& xpark) {
    const int bnxt = bcur == 2 ? 0 : bcur + 1;
    if constexpr (!(ABL & ABL_NOX)) load_x(t + 2, xnew);
    if constexpr (!(ABL & ABL_NOW)) load_w(t + 3, wnew);
    SB;
    const uint32_t z = ((wcur.z >> zshift) & 15u) + ZOFF;
    const f16x2 zc1 = as_h2((z * 0x00010001u) | 0xE400E400u);
    const f16x2 zc2 = as_h2((z * 0x00100010u) | 0xD400D400u);
    const f16x2 sc2 = as_h2(__builtin_amdgcn_perm(wcur.s, wcur.s, sshift ? 0x03020302u : 0x01000100u));
    auto dequant = [&](int s) {
      const uint32_t w = wcur.w[s], w8 = w >> 8;
      u32x4 bf;
      if constexpr (ABL & ABL_NODEQ) {
        bf = u32x4{w, w8, w, w8};
      } else {
        const f16x2 d0 = as_h2((w & 0x000f000fu) | 0x64006400u) + zc1;
        const f16x2 d1 = __builtin_elementwise_fma(as_h2((w & 0x00f000f0u) | 0x64006400u), sixteenth, zc2);
        const f16x2 d2 = as_h2((w8 & 0x000f000fu) | 0x64006400u) + zc1;
        const f16x2 d3 = __builtin_elementwise_fma(as_h2((w8 & 0x00f000f0u) | 0x64006400u), sixteenth, zc2);
        bf[0] = as_u32(d0 * sc2); bf[1] = as_u32(d1 * sc2);
        bf[2] = as_u32(d2 * sc2); bf[3] = as_u32(d3 * sc2);
      }
      return bf;
    };
    u32x4 b0 = dequant(0);
    read_a(bcur, 1, fb);
    SB;
    mma(fa, b0);
    u32x4 b1 = dequant(1);
    SB;
    if constexpr (!(ABL & ABL_NOX)) store_x(bnxt, xpark);   // x(t+1); past the end it parks a re-read tile
    read_a(bcur, 2, fa);
    SB;
    mma(fb, b1);
    b0 = dequant(2);
    SB;
    __syncthreads();             // every wave has parked its part of x(t+1)
    read_a(bcur, 3, fb);
    SB;
    mma(fa, b0);
    b1 = dequant(3);
    SB;
    read_a(bnxt, 0, fa);         // first fragments of tile t+1
    SB;
    mma(fb, b1);
    bcur = bnxt;
  };
#undef SB

  WStep w0, w1, w2, w3;
  XSet xa, xb_;
  load_x(0, xa);
  load_x(1, xb_);
  load_w(0, w0);
  load_w(1, w1);
  load_w(2, w2);
  w3 = w2;
  store_x(0, xa);
  if constexpr (ABL & ABL_NOX) { store_x(1, xb_); store_x(2, xb_); }
  __syncthreads();
  read_a(0, 0, fa);

  int t = 0;
  for (; t + 4 <= nsteps; t += 4) {
    k_step(t, w0, w3, xa, xb_);
    k_step(t + 1, w1, w0, xb_, xa);
    k_step(t + 2, w2, w1, xa, xb_);
    k_step(t + 3, w3, w2, xb_, xa);
  }
  if (t < nsteps) k_step(t, w0, w3, xa, xb_);
  if (t + 1 < nsteps) k_step(t + 1, w1, w0, xb_, xa);
  if (t + 2 < nsteps) k_step(t + 2, w2, w1, xa, xb_);

  if (!nok) return;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + 128 * wm + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row >= m) continue;
      outp[(int64_t)row * n + ncol] = (_Float16)acc[a][i];
    }
}

template <int ABL>
static float run3(const char* name, int m, int n, int k, _Float16* x, std::vector<uint32_t*>& qw, uint32_t* qz,
                  _Float16* sc, _Float16* out, int iters) {
  dim3 grid((n + 127) / 128, (m + 255) / 256, 1);
  CK(hipFuncSetAttribute((const void*)q4_kernel3<0, 4, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * X_TILE_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((q4_kernel3<0, 4, ABL>), grid, dim3(512), 3 * X_TILE_BYTES, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((q4_kernel3<0, 4, ABL>), grid, dim3(512), 3 * X_TILE_BYTES, 0, x, qw[i % qw.size()], qz, sc, out, m, n, k, 128);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const float us = ms * 1e3f / iters;
  printf("%-28s n=%5d k=%5d  %8.1f us  %7.1f TFLOP/s\n", name, n, k, us, 2.0 * m * n * k / (us * 1e-6) / 1e12);
  return us;
}


int main(int argc, char** argv) {
  const int m = 256;
  int n = 28672, k = 4096;
  if (argc > 2) { n = atoi(argv[1]); k = atoi(argv[2]); }
  const int iters = 20, sets = 6;
  _Float16 *x, *sc, *out, *out_ref; uint32_t* qz; std::vector<uint32_t*> qw(sets);
  CK(hipMalloc(&x, (size_t)m * k * 2)); CK(hipMalloc(&sc, (size_t)(k / 128) * n * 2));
  CK(hipMalloc(&out, (size_t)m * n * 2)); CK(hipMalloc(&out_ref, (size_t)m * n * 2));
  CK(hipMalloc(&qz, (size_t)(k / 128) * (n / 8) * 4));
  std::vector<uint32_t> hw((size_t)(k / 8) * n);
  srand(1);
  for (int s = 0; s < sets; ++s) {
    for (auto& v : hw) v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    CK(hipMalloc(&qw[s], hw.size() * 4));
    CK(hipMemcpy(qw[s], hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  }
  std::vector<uint32_t> hz((size_t)(k / 128) * (n / 8));
  for (auto& v : hz) v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  CK(hipMemcpy(qz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
  std::vector<_Float16> hs((size_t)(k / 128) * n), hx((size_t)m * k);
  for (auto& v : hs) v = (_Float16)(0.005f + 0.01f * (rand() % 1000) / 1000.f);
  for (auto& v : hx) v = (_Float16)((rand() % 2001 - 1000) / 1000.f);
  CK(hipMemcpy(sc, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));

  if (argc > 3 && !strcmp(argv[3], "base")) { run<0, 0>("baseline (pointer loads)", m, n, k, x, qw, qz, sc, out_ref, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v5")) { run5<0, 0>("V5", m, n, k, x, qw, qz, sc, out, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v5wide")) { run5<0, 16>("V5 wide w", m, n, k, x, qw, qz, sc, out, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v5widenozs")) { run5<0, 48>("V5 wide w, no z/s", m, n, k, x, qw, qz, sc, out, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v5widenox")) { run5<ABL_NOX, 16>("V5 wide w, no x", m, n, k, x, qw, qz, sc, out, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v5nox")) { run5<ABL_NOX, 0>("V5 nox", m, n, k, x, qw, qz, sc, out, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v5now")) { run5<ABL_NOW, 0>("V5 now", m, n, k, x, qw, qz, sc, out, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v5noxw")) { run5<ABL_NOX | ABL_NOW, 0>("V5 noxw", m, n, k, x, qw, qz, sc, out, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "mfma")) { run<ABL_NOX | ABL_NOW | ABL_NODEQ | ABL_NOLDSREAD, 1>("mfma only", m, n, k, x, qw, qz, sc, out, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v3nodq")) { run3<ABL_NODEQ>("V3 nodq", m, n, k, x, qw, qz, sc, out_ref, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v3")) { run3<0>("V3", m, n, k, x, qw, qz, sc, out_ref, 5); return 0; }
  if (argc > 3 && !strcmp(argv[3], "v2")) { run2<0>("V2", m, n, k, x, qw, qz, sc, out_ref, 5); return 0; }
  run<0, 0>("baseline (pointer loads)", m, n, k, x, qw, qz, sc, out_ref, iters);
  run<0, 1>("buffer loads + and_or", m, n, k, x, qw, qz, sc, out, iters);
  // compare V1 against V0 on the same weight set (last launch of each used set (iters-1) % sets)
  {
    std::vector<_Float16> a((size_t)m * n), b((size_t)m * n);
    CK(hipMemcpy(a.data(), out_ref, a.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out, b.size() * 2, hipMemcpyDeviceToHost));
    double md = 0; for (size_t i = 0; i < a.size(); ++i) { double d = fabs((double)a[i] - (double)b[i]); if (d > md) md = d; }
    printf("max |V1 - V0| = %g\n", md);
  }
  run2<0>("V2 deep prefetch", m, n, k, x, qw, qz, sc, out, iters);
  {
    std::vector<_Float16> a((size_t)m * n), b((size_t)m * n);
    CK(hipMemcpy(a.data(), out_ref, a.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out, b.size() * 2, hipMemcpyDeviceToHost));
    double md = 0; for (size_t i = 0; i < a.size(); ++i) { double d = fabs((double)a[i] - (double)b[i]); if (d > md) md = d; }
    printf("max |V2 - V0| = %g\n", md);
  }
  run5<0, 0>("V5 128x256", m, n, k, x, qw, qz, sc, out, iters);
  {
    std::vector<_Float16> a((size_t)m * n), b((size_t)m * n);
    CK(hipMemcpy(a.data(), out_ref, a.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out, b.size() * 2, hipMemcpyDeviceToHost));
    double md = 0; for (size_t i = 0; i < a.size(); ++i) { double d = fabs((double)a[i] - (double)b[i]); if (d > md) md = d; }
    printf("max |V5 - V0| = %g\n", md);
  }
  run5<0, 8>("V5 setprio", m, n, k, x, qw, qz, sc, out, iters);
  run5<0, 2>("V5 andor", m, n, k, x, qw, qz, sc, out, iters);
  run5<0, 10>("V5 setprio+andor", m, n, k, x, qw, qz, sc, out, iters);
  return 0;
  run5<ABL_NODEQ, 0>("V5 no dequant", m, n, k, x, qw, qz, sc, out, iters);
  run5<ABL_NOX, 0>("V5 no x staging", m, n, k, x, qw, qz, sc, out, iters);
  run5<ABL_NOW, 0>("V5 no w loads", m, n, k, x, qw, qz, sc, out, iters);
  run5<ABL_NOW | ABL_NOX, 0>("V5 no x no w", m, n, k, x, qw, qz, sc, out, iters);
  run4<0, 0>("V4 opt0", m, n, k, x, qw, qz, sc, out, iters);
  run4<0, 1>("V4 saddr", m, n, k, x, qw, qz, sc, out, iters);
  run4<0, 2>("V4 andor", m, n, k, x, qw, qz, sc, out, iters);
  run4<0, 3>("V4 saddr+andor", m, n, k, x, qw, qz, sc, out, iters);
  run4<0, 4>("V4 sgb", m, n, k, x, qw, qz, sc, out, iters);
  run4<0, 7>("V4 all", m, n, k, x, qw, qz, sc, out, iters);
  {
    std::vector<_Float16> a((size_t)m * n), b((size_t)m * n);
    CK(hipMemcpy(a.data(), out_ref, a.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out, b.size() * 2, hipMemcpyDeviceToHost));
    double md = 0; for (size_t i = 0; i < a.size(); ++i) { double d = fabs((double)a[i] - (double)b[i]); if (d > md) md = d; }
    printf("max |V4 - V0| = %g\n", md);
  }
  return 0;
  run3<0>("V3 mid-step barrier", m, n, k, x, qw, qz, sc, out, iters);
  {
    std::vector<_Float16> a((size_t)m * n), b((size_t)m * n);
    CK(hipMemcpy(a.data(), out_ref, a.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out, b.size() * 2, hipMemcpyDeviceToHost));
    double md = 0; for (size_t i = 0; i < a.size(); ++i) { double d = fabs((double)a[i] - (double)b[i]); if (d > md) md = d; }
    printf("max |V3 - V0| = %g\n", md);
  }
  run3<ABL_NODEQ>("V3 no dequant", m, n, k, x, qw, qz, sc, out, iters);
  run3<ABL_NOX>("V3 no x staging", m, n, k, x, qw, qz, sc, out, iters);
  run3<ABL_NOW>("V3 no w loads", m, n, k, x, qw, qz, sc, out, iters);
  run2<ABL_NODEQ>("V2 no dequant", m, n, k, x, qw, qz, sc, out, iters);
  run2<ABL_NOX>("V2 no x staging", m, n, k, x, qw, qz, sc, out, iters);
  run2<ABL_NOW>("V2 no w loads", m, n, k, x, qw, qz, sc, out, iters);
  run<ABL_NODEQ, 1>("V1 no dequant", m, n, k, x, qw, qz, sc, out, iters);
  run<ABL_NOX, 1>("V1 no x staging", m, n, k, x, qw, qz, sc, out, iters);
  run<ABL_NOW, 1>("V1 no w loads", m, n, k, x, qw, qz, sc, out, iters);
  run<ABL_NOMFMA, 1>("V1 no mfma", m, n, k, x, qw, qz, sc, out, iters);
  run<ABL_NOLDSREAD, 1>("V1 no lds reads", m, n, k, x, qw, qz, sc, out, iters);
  run<ABL_NOX | ABL_NOW, 1>("V1 no x, no w", m, n, k, x, qw, qz, sc, out, iters);
  run<ABL_NOX | ABL_NOW | ABL_NODEQ, 1>("V1 no x, no w, no deq", m, n, k, x, qw, qz, sc, out, iters);
  run<ABL_NOX | ABL_NOW | ABL_NODEQ | ABL_NOLDSREAD, 1>("V1 mfma only", m, n, k, x, qw, qz, sc, out, iters);
  return 0;
}
