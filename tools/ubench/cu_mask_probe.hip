// cu_mask_probe.hip -- which physical CUs does bit i of a hipExtStreamCreateWithCUMask mask select on this device?
// For every 32-bit word w of the mask (and a few single bits) a kernel of 2048 workgroups runs on a stream masked to that word and records
// (XCC_ID, SE, CU) of every workgroup; prints the set of (xcc, se, cu) each mask reached.  Build: hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void probe(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
  }
}
int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  printf("CUs: %d\n", ncu);
  const int NW = (ncu + 31) / 32, NB = 2048;
  unsigned* d; hipMalloc(&d, NB * 8);
  std::vector<unsigned> h(NB * 2);
  auto run = [&](const std::vector<unsigned>& mask, const char* name) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    hipMemsetAsync(d, 0xff, NB * 8, s);
    hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per;     // xcc -> set of (se, sh, cu)
    for (int i = 0; i < NB; ++i) { const unsigned hw = h[2 * i], x = h[2 * i + 1] & 0xf; per[x].insert((hw >> 8) & 0xfff); }
    printf("%s:", name);
    int tot = 0;
    for (auto& kv : per) { printf("  xcc%u: %zu CUs", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  | total %d\n", tot);
    hipStreamDestroy(s);
  };
  for (int w = 0; w < NW; ++w) { std::vector<unsigned> m(NW, 0u); m[w] = 0xffffffffu; char nm[32]; snprintf(nm, 32, "word %d", w); run(m, nm); }
  for (int b : {0, 1, 2, 7, 8, 9, 31, 32}) { std::vector<unsigned> m(NW, 0u); m[b / 32] = 1u << (b % 32); char nm[32]; snprintf(nm, 32, "bit %d", b); run(m, nm); }
  { std::vector<unsigned> m(NW, 0u); for (int i = 0; i < ncu; i += 8) m[i / 32] |= 1u << (i % 32); run(m, "bits 0,8,16,.."); }
  { std::vector<unsigned> m(NW, 0x01010101u); run(m, "every 8th bit"); }
  return 0;
}
