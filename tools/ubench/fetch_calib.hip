// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns the
// conv / wgrad kernels use: a known byte count (well beyond the 256 MiB
// Infinity Cache) is read once with
//   k_read_b32   one dword per lane, 64 lanes = 256 contiguous bytes per instruction, rows of 7680 floats
//                (conv_gemm_x3_kernel's activation fetch: buffer_load_dword, lane = column)
//   k_read_b128  16 bytes per lane (the guide's calibrated case; weights, wgrad operands)
//   k_buf_b32    the same dword pattern through a buffer descriptor with a scalar offset
// and written once with k_write_b32 (the MFMA C-layout epilogue: 32 lanes x 4 B per row, two rows per
// instruction).  Run under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`; the factor
// is bytes / (counter KiB x 1024).  Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__global__ __launch_bounds__(256) void k_read_b32(const float* __restrict__ x, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i];
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read_b128(const float4* __restrict__ x, size_t n4, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = x[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_buf_b32(const float* x, size_t n, float* out) {
  // one descriptor per 1 GiB window; per-thread dword offset + scalar offset per 256-float step
  float acc = 0.f;
  const size_t per = (size_t)1 << 28;                       // floats per window
  for (size_t w0 = 0; w0 < n; w0 += per) {
    const size_t len = n - w0 < per ? n - w0 : per;
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + w0), 0, (int)(len * 4), 0x00020000);
    for (size_t i = (size_t)blockIdx.x * 256; i < len; i += (size_t)gridDim.x * 256)
      acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 4u * threadIdx.x, (unsigned)(4 * i), 0));
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write_b32(float* __restrict__ y, int rows, int T) {
  // C-layout store pattern: a wave writes 32 consecutive floats of row r and of row r + 4 per instruction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
  const int tiles_t = T / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < (long)(rows / 8) * tiles_t; tile += (long)gridDim.x * 4) {
    const long r0 = (tile / tiles_t) * 8, t0 = (tile % tiles_t) * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[(r0 + r + 4 * lk) * T + t0 + li] = (float)r;
  }
}

int main(int argc, char** argv) {
  const size_t rows = 16 * 256 * 4, T = 7680;               // 4 x the (B, C, T) activation: 503 MB
  const size_t n = rows * T;
  float *x, *out;
  hipMalloc(&x, n * 4); hipMalloc(&out, 4);
  hipMemset(x, 0, n * 4);
  hipDeviceSynchronize();
  const int grid = 256 * 8;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_read_b32, dim3(grid), dim3(256), 0, 0, x, n, out);
    hipLaunchKernelGGL(k_read_b128, dim3(grid), dim3(256), 0, 0, (const float4*)x, n / 4, out);
    hipLaunchKernelGGL(k_buf_b32, dim3(grid), dim3(256), 0, 0, x, n, out);
    hipLaunchKernelGGL(k_write_b32, dim3(grid), dim3(256), 0, 0, x, (int)rows, (int)T);
    hipDeviceSynchronize();
  }
  printf("bytes per launch: %zu (%.1f KiB)\n", n * 4, n * 4 / 1024.0);
  return 0;
}
