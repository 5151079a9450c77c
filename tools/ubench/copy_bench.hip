// The achievable HBM -> HBM copy line on MI355X, by hand (Next 4): what RCAS — 8 B read + 8 B written per pixel —
// is measured against.  MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; rounds 1-4 measured 5.0-5.4 TB/s with torch's
// elementwise copy of one 4K image (a 25 us launch with its own ramp and tail).  This program measures, on >= 1 GiB per buffer
// (four times the 256 MB Infinity Cache: nothing is served from it):
//   1. grid-stride float4 copies: plain, non-temporal stores, non-temporal loads + stores; 16 and 32 bytes per lane per iteration;
//      grids of 2 048 ... 65 536 workgroups of 256 threads
//   2. the RCAS kernel's own access pattern WITHOUT its arithmetic: an image batch (N x 3840 x 2160 RGBA16F) walked in
//      128-column x R-row strips per wave, one 16-byte (or 32-byte: 256 columns) load and store per lane and row, rows K deep in
//      flight, the XCD-contiguous strip order of the product kernel — so that "pattern" and "arithmetic" can be told apart
//   3. a single 4K image (66 MB) the same ways: the launch-ramp-and-tail cost at the size of one frame
// Build: hipcc --offload-arch=gfx950 -O3 copy_bench.hip -o copy_bench     Run: ./copy_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT_LOAD, bool NT_STORE>
__device__ __forceinline__ void mov(f4* dst, const f4* src) {
  const f4 v = NT_LOAD ? __builtin_nontemporal_load(src) : *src;
  if (NT_STORE) __builtin_nontemporal_store(v, dst); else *dst = v;
}

// grid-stride copy, PER x 16 bytes per lane per iteration (the PER loads are issued before the stores)
template <int PER, bool NT_LOAD, bool NT_STORE>
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * PER;
  for (size_t i = (size_t)blockIdx.x * 256 * PER + threadIdx.x; i < n; i += stride) {
    f4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = i + (size_t)k * 256 < n ? (NT_LOAD ? __builtin_nontemporal_load(in + i + (size_t)k * 256) : in[i + (size_t)k * 256]) : f4{};
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (i + (size_t)k * 256 < n) { if (NT_STORE) __builtin_nontemporal_store(v[k], out + i + (size_t)k * 256); else out[i + (size_t)k * 256] = v[k]; }
  }
}

__device__ __forceinline__ int xcd_swizzle(int b, int n) {
  const int q = n / 8, r = n % 8, xcd = b % 8, idx = b / 8;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// RCAS's access pattern: image rows `pitch` bytes apart; a wave owns LANE_BYTES x 64 bytes of a row x `rows` rows and walks down,
// DEPTH rows in flight; two waves per workgroup side by side.  LANE_BYTES = 16: 128 RGBA16F columns per wave (the product kernel);
// 32: 256 columns per wave.
template <int LANE_BYTES, int DEPTH, bool NT_STORE>
__global__ void __launch_bounds__(128) k_strips(const char* __restrict__ in, char* __restrict__ out, int width_bytes, int height, long long pitch,
                                                 long long frame_stride, int tiles_x, int tiles_y, int frames, int rows) {
  constexpr int V = LANE_BYTES / 16;
  const int per_frame = tiles_x * tiles_y;
  const int t = xcd_swizzle(blockIdx.x, per_frame * frames);
  const int frame = t / per_frame, tf = t - frame * per_frame;
  const int ty = tf / tiles_x, tx = tf - ty * tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = (tx * 2 + wave) * 64 * LANE_BYTES + lane * LANE_BYTES;
  if (x >= width_bytes) return;
  const int y0 = ty * rows, y1 = min(y0 + rows, height);
  const char* src = in + frame * frame_stride + x;
  char* dst = out + frame * frame_stride + x;
  f4 q[DEPTH][V];
#pragma unroll
  for (int k = 0; k < DEPTH - 1; ++k)
#pragma unroll
    for (int v = 0; v < V; ++v) q[k][v] = *reinterpret_cast<const f4*>(src + (long long)min(y0 + k, y1 - 1) * pitch + 16 * v);
#pragma unroll 1
  for (int y = y0; y < y1; y += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
#pragma unroll
      for (int v = 0; v < V; ++v) q[(k + DEPTH - 1) % DEPTH][v] = *reinterpret_cast<const f4*>(src + (long long)min(y + k + DEPTH - 1, y1 - 1) * pitch + 16 * v);
      if (y + k < y1) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          f4* d = reinterpret_cast<f4*>(dst + (long long)(y + k) * pitch + 16 * v);
          if (NT_STORE) __builtin_nontemporal_store(q[k][v], d); else *d = q[k][v];
        }
      }
    }
  }
}

template <class F>
static float time_us(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int blk = 0; blk < 5; ++blk) {
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    t.push_back(ms * 1e3f / reps);
  }
  std::sort(t.begin(), t.end());
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return t[t.size() / 2];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d\n", p.name, p.multiProcessorCount);
  // ---- 1. grid-stride copies over 2 GiB buffers, rotating over two buffer pairs ----
  const size_t bytes = 2048ull << 20, n = bytes / 16;
  f4 *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  // warm clocks
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_copy<1, false, false>), dim3(8192), dim3(256), 0, 0, a, b, n);
  CK(hipDeviceSynchronize());
#define COPY(PER, NL, NS, G) { const float us = time_us([&] { hipLaunchKernelGGL((k_copy<PER, NL, NS>), dim3(G), dim3(256), 0, 0, a, b, n); }, 4); \
    printf("copy 2 GiB  %2d B/lane  nt_load=%d nt_store=%d  grid %6d: %8.1f us  %.2f TB/s (read + write)\n", 16 * PER, NL, NS, G, us, 2.0 * bytes / us * 1e-6); }
  for (int g : {2048, 4096, 8192, 16384, 65536}) {
    COPY(1, false, false, g) COPY(1, false, true, g) COPY(1, true, true, g)
    COPY(2, false, false, g) COPY(2, false, true, g) COPY(2, true, true, g)
    COPY(4, false, true, g)
  }
  // ---- 2. / 3. RCAS's strip pattern on 3840 x 2160 RGBA16F images ----
  const int W = 3840, H = 2160;
  const long long pitch = (long long)W * 8, fstride = pitch * H;
  for (int frames : {1, 8, 16}) {
    if ((size_t)frames * fstride > bytes) continue;
    printf("-- %d x %dx%d RGBA16F (%.0f MB read + %.0f MB written)%s\n", frames, W, H, frames * fstride / 1e6, frames * fstride / 1e6,
           frames == 1 ? "  [input rotates over the 2 GiB buffer: read from HBM]" : "");
    const size_t slots = frames == 1 ? bytes / fstride : 1;
    size_t slot = 0;
#define STRIPS(LB, D, NS, ROWS) { const int tx = (W * 8 + 128 * LB - 1) / (128 * LB), ty = (H + ROWS - 1) / ROWS; \
      const float us = time_us([&] { const size_t off = (slot++ % slots) * fstride; \
        hipLaunchKernelGGL((k_strips<LB, D, NS>), dim3(tx * ty * frames), dim3(128), 0, 0, (const char*)a + off, (char*)b + off, W * 8, H, pitch, fstride, tx, ty, frames, ROWS); }, frames == 1 ? 40 : 6); \
      printf("strips %2d B/lane depth %d nt_store=%d rows %3d (%6d wgs): %8.1f us  %.2f TB/s\n", LB, D, NS, ROWS, tx * ty * frames, us, 2.0 * frames * fstride / us * 1e-6); }
    for (int rows : {8, 16, 32, 64}) {
      if (rows == 8)  { STRIPS(16, 2, true, 8) STRIPS(16, 8, true, 8) STRIPS(32, 2, true, 8) STRIPS(32, 4, true, 8) STRIPS(16, 8, false, 8) }
      if (rows == 16) { STRIPS(16, 2, true, 16) STRIPS(16, 8, true, 16) STRIPS(32, 2, true, 16) STRIPS(32, 4, true, 16) STRIPS(16, 8, false, 16) }
      if (rows == 32) { STRIPS(16, 2, true, 32) STRIPS(16, 8, true, 32) STRIPS(32, 2, true, 32) STRIPS(32, 4, true, 32) STRIPS(16, 8, false, 32) }
      if (rows == 64) { STRIPS(16, 8, true, 64) STRIPS(32, 4, true, 64) }
    }
    // the same bytes as a flat grid-stride copy
    const size_t nn = (size_t)frames * fstride / 16;
    for (int g : {2048, 8192}) {
      const float us = time_us([&] { const size_t off = (slot++ % slots) * fstride / 16; hipLaunchKernelGGL((k_copy<1, false, true>), dim3(g), dim3(256), 0, 0, a + off, b + off, nn); }, frames == 1 ? 40 : 6);
      printf("flat copy 16 B/lane nt_store grid %5d: %8.1f us  %.2f TB/s\n", g, us, 2.0 * nn * 16 / us * 1e-6);
    }
  }
  return 0;
}
