// Where do the workgroups of a stream created with hipExtStreamCreateWithCUMask run on MI355X (8 XCDs x 32 CUs), and what does
// a mask cost a chip-filling kernel?  Prints, per mask, the CUs used per XCD and the time of a fixed amount of spin work.
// build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <map>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void where_kernel(unsigned* out, int spin) {
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
  unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = (xcc & 0xf) | (a == 123.f ? 16u : 0u); }
}
static int run(const char* name, const std::vector<uint32_t>& mask, unsigned* dout, int nblk) {
  hipStream_t st;
  if (mask.empty()) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  else CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, st, dout, 20000);      // warm
  CK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, st, dout, 20000);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> h(2 * nblk);
  CK(hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost));
  std::map<int, std::set<int>> per;
  for (int i = 0; i < nblk; ++i) {
    const unsigned hw = h[2 * i], x = h[2 * i + 1] & 0xf;
    const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per[x].insert(se * 32 + sh * 16 + cu);
  }
  int tot = 0;
  printf("%-28s %7.3f ms  CUs per XCD:", name, ms);
  for (auto& p : per) { printf(" x%d:%zu", p.first, p.second.size()); tot += (int)p.second.size(); }
  printf("  total %d\n", tot);
  CK(hipStreamDestroy(st));
  return 0;
}
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("multiProcessorCount %d\n", pr.multiProcessorCount);
  const int nblk = 256 * 8 * 4;
  unsigned* dout; CK(hipMalloc(&dout, 2 * nblk * 4));
  auto bits = [](int lo, int hi, int step = 1) { std::vector<uint32_t> m(8, 0u); for (int i = lo; i < hi; i += step) m[i / 32] |= 1u << (i % 32); return m; };
  if (run("no mask", {}, dout, nblk)) return 1;
  if (run("bits 0..255", bits(0, 256), dout, nblk)) return 1;
  if (run("bits 0..127", bits(0, 128), dout, nblk)) return 1;
  if (run("bits 128..255", bits(128, 256), dout, nblk)) return 1;
  if (run("bits 0..223", bits(0, 224), dout, nblk)) return 1;
  if (run("bits 0..191", bits(0, 192), dout, nblk)) return 1;
  if (run("even bits", bits(0, 256, 2), dout, nblk)) return 1;
  if (run("bits 0..31", bits(0, 32), dout, nblk)) return 1;
  if (run("bits 0..7", bits(0, 8), dout, nblk)) return 1;
  { std::vector<uint32_t> m(8, 0u); for (int i = 0; i < 256; ++i) if ((i % 8) != 7) m[i / 32] |= 1u << (i % 32); if (run("all but every 8th", m, dout, nblk)) return 1; }
  { std::vector<uint32_t> m(8, 0u); for (int i = 0; i < 256; ++i) if ((i / 8) % 4 != 3) m[i / 32] |= 1u << (i % 32); if (run("all but bits 24..31 mod 32", m, dout, nblk)) return 1; }
  return 0;
}
