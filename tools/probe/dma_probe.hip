// Hardware probe (GPU box): how fast can a CU ingest a K/V-like tile stream, and does the path matter?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/dma_probe.hip -o tools/probe/dma_probe && tools/probe/dma_probe
// Every workgroup (256 threads) streams `tiles` 16-KiB tiles (128 rows of 128 B at a 1536-B row stride: the OPT-125m
// K / V layout; a K tile + a V tile of the product kernels are the same 16 KiB) of its own (batch, head) with `depth` tiles in flight:
//   mode 0  global_load_lds_dwordx4 (LDS-DMA, the product kernels' path), 4 pieces of 1 KiB per wave and tile
//   mode 1  global_load_dwordx4 into registers, then ds_write_b128 (register-staged)
//   mode 2  global_load_dwordx4 into registers only (no LDS write)
// `share` workgroups read the SAME head (as the q tiles of one head do: 1 = every byte from HBM, 4 = 3 of 4 are L2 hits).
// Reported: microseconds per launch, bytes requested by the CUs per second, unique bytes per second.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned char* __restrict__ src, unsigned* __restrict__ sink, int tiles, int share, int heads, long head_stride,
                                                     long batch_stride, int row_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[DEPTH * 16384];
  const int bid = blockIdx.x;
  const int nbh = gridDim.x / share;
  const int bh = bid % nbh;  // sharers of a (batch, head) are nbh = 192 block ids apart: the same XCD (block id mod 8), as in the product kernels
  const int b = bh / heads, h = bh % heads;
  const unsigned char* base = src + (long)b * batch_stride + (long)h * head_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;
  // piece j of wave w: rows (4 w + j) * 8 .. +7 of the tile, lane -> (row lane / 8, 16-B chunk lane % 8)
  unsigned off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) off[j] = (unsigned)(((wave * 4 + j) * 8 + lane / 8) * row_stride + (lane % 8) * 16);
  const long tile_step = 128L * row_stride;
  u4 acc = {0u, 0u, 0u, 0u};
  if constexpr (MODE == 0) {
    int issued = 0;
    const unsigned char* cur = base;
    for (; issued < DEPTH && issued < tiles; ++issued) {
      const unsigned slot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(issued % DEPTH) * 16384u + (unsigned)wave * 4096u);
#pragma unroll
      for (int j = 0; j < 4; ++j) glds16_s(cur, off[j], slot + j * 1024);
      cur += tile_step;
    }
    for (int i = 0; i < tiles; ++i) {
      // tile i landed (this wave's pieces): at most (issued - i - 1) tiles = 4x pieces may stay in flight
      if (issued - i - 1 >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (issued - i - 1 == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (issued < tiles) {
        const unsigned slot = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(issued % DEPTH) * 16384u + (unsigned)wave * 4096u);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_s(cur, off[j], slot + j * 1024);
        cur += tile_step;
        ++issued;
      }
    }
    acc.x = *reinterpret_cast<const unsigned*>(lds + tid * 16);
  } else {
    static_assert(DEPTH <= 3, "register stages");
    u4 st[DEPTH][4];
    const unsigned char* cur = base;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (d < tiles) {
#pragma unroll
        for (int j = 0; j < 4; ++j) st[d][j] = *reinterpret_cast<const u4*>(cur + off[j]);
        cur += tile_step;
      }
    }
    for (int i0 = 0; i0 < tiles; i0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int i = i0 + d;
        if (i < tiles) {
          if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<u4*>(lds + (d * 16384 + wave * 4096 + j * 1024 + lane * 16)) = st[d][j];
            __syncthreads();
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc ^= st[d][j];
          }
          if (i + DEPTH < tiles) {
#pragma unroll
            for (int j = 0; j < 4; ++j) st[d][j] = *reinterpret_cast<const u4*>(cur + off[j]);
            cur += tile_step;
          }
        }
      }
    }
    if constexpr (MODE == 1) acc.x = *reinterpret_cast<const unsigned*>(lds + tid * 16);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[bid] = acc.x;  // keep the loads alive
}

template <int MODE, int DEPTH>
static float run(const unsigned char* src, unsigned* sink, int wgs, int tiles, int share, int heads, long hs, long bs, int rs, int nbuf, size_t buf_bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) stream_kernel<MODE, DEPTH><<<wgs, 256>>>(src + (size_t)(i % nbuf) * buf_bytes, sink, tiles, share, heads, hs, bs, rs);
  if (hipDeviceSynchronize() != hipSuccess) { std::printf("mode %d depth %d: launch failed\n", MODE, DEPTH); std::exit(1); }
  const int iters = 200;
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) stream_kernel<MODE, DEPTH><<<wgs, 256>>>(src + (size_t)(i % nbuf) * buf_bytes, sink, tiles, share, heads, hs, bs, rs);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  // OPT-125m K and V as one tensor of 1024 rows per batch: B=16, H=12, d=64 fp16: rows of 128 B, row stride 1536 B, head stride 128 B
  const int B = 16, S = 1024, H = 12;
  const int rs = 1536;
  const long hs = 128, bs = (long)S * rs;
  const size_t buf_bytes = (size_t)B * bs;  // 25.2 MB
  const int nbuf = 20;                       // 503 MB > Infinity Cache
  unsigned char* src;
  unsigned* sink;
  if (hipMalloc(&src, buf_bytes * nbuf) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, buf_bytes * nbuf);
  std::printf("src %p .. %p\n", (void*)src, (void*)(src + buf_bytes * nbuf));
  std::fflush(stdout);
  std::printf("%-44s %9s %14s %14s\n", "stream", "us", "CU-side GB/s", "unique GB/s");
  for (int share : {1, 2, 4}) {
    const int tiles = 8;  // 8 tiles of 128 rows = one head's K and V
    const int wgs = B * H * share;
    const double cu_bytes = (double)wgs * tiles * 16384, uniq = (double)B * H * tiles * 16384;
    float t;
    std::fflush(stdout);
    t = run<0, 2>(src, sink, wgs, tiles, share, H, hs, bs, rs, nbuf, buf_bytes);
    std::fflush(stdout); std::printf("lds-dma  depth 2  share %d  (%4d WGs)          %9.2f %14.1f %14.1f\n", share, wgs, t, cu_bytes / t / 1e3, uniq / t / 1e3);
    t = run<0, 3>(src, sink, wgs, tiles, share, H, hs, bs, rs, nbuf, buf_bytes);
    std::printf("lds-dma  depth 3  share %d  (%4d WGs)          %9.2f %14.1f %14.1f\n", share, wgs, t, cu_bytes / t / 1e3, uniq / t / 1e3);
    t = run<1, 2>(src, sink, wgs, tiles, share, H, hs, bs, rs, nbuf, buf_bytes);
    std::printf("regs+ds_write depth 2  share %d  (%4d WGs)     %9.2f %14.1f %14.1f\n", share, wgs, t, cu_bytes / t / 1e3, uniq / t / 1e3);
    t = run<2, 2>(src, sink, wgs, tiles, share, H, hs, bs, rs, nbuf, buf_bytes);
    std::printf("regs only depth 2  share %d  (%4d WGs)         %9.2f %14.1f %14.1f\n", share, wgs, t, cu_bytes / t / 1e3, uniq / t / 1e3);
    t = run<2, 3>(src, sink, wgs, tiles, share, H, hs, bs, rs, nbuf, buf_bytes);
    std::printf("regs only depth 3  share %d  (%4d WGs)         %9.2f %14.1f %14.1f\n", share, wgs, t, cu_bytes / t / 1e3, uniq / t / 1e3);
  }
  return 0;
}
