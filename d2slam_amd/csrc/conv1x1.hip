// conv1x1.hip -- the detector head's last layer, convPb: 1x1, 256 -> 65 channels, no activation (fp32 modes).
//
// Replaces the TensorRT engine execution at d2frontend/src/CNN/superpoint_tensorrt.cpp:150 for that layer (network:
// d2frontend/superpoint.ipynb:300-374).  Same arithmetic as the generic kernels (conv.hip / conv_pc.hip): every output is ONE fmaf chain
// from the bias over the input channels in ascending order -- the matrix pipe's v_mfma_f32_32x32x2_f32 is that chain for the 64 cell
// channels, and the 65th channel (the dustbin) is the same chain on the vector pipe, so that the padded third 32-channel tile of the
// generic kernels (1 real channel of 32: a third of their matrix-pipe work) disappears.  Bit-identical to them and to the oracle.
//
// A wave owns 32 consecutive pixels of the flat pixel list (all images): the A operand (pixel = lane & 31, k = 2 step + (lane >> 5))
// comes straight from HBM as two float4 per 8 input channels and lane (a ring of 8 such groups in flight), the B fragments of the 64
// cell channels sit in LDS for the life of the persistent workgroup in the packed order of pack_weights_f32 (one ds_read_b128 per 32
// channels and four k-steps), the dustbin weights as two broadcast LDS reads per 8 channels.
#include "conv_common.h"

namespace d2fe {

namespace {
constexpr int C1_CIN = 256, C1_C8 = C1_CIN / 8, C1_RING = 8;
}

__global__ __launch_bounds__(256, 2) void conv1x1_256_65_kernel(ConvArgs a, long npix) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];      // [2 n-tiles][32 groups of 8 channels][64 lanes][4 k-steps]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const float* wp = reinterpret_cast<const float*>(a.wpack);
  for (int i = tid; i < 2 * C1_C8 * 64; i += 256)
    reinterpret_cast<f32x4*>(wsm)[i] = reinterpret_cast<const f32x4*>(wp)[i];
  // the dustbin (channel 64) is lane 0 (k even) / lane 32 (k odd) of the third n-tile's records (k = 8 g + 2 q + h at [g][h * 32][q]):
  // kept in LDS as [g][even k 4 | odd k 4] and read as two broadcast float4 per group
  float* wdl = wsm + 2 * C1_C8 * 256;
  if (tid < C1_C8 * 8) {
    const int g = tid >> 3, h = (tid >> 2) & 1, q = tid & 3;
    wdl[tid] = wp[((size_t)2 * C1_C8 + g) * 256 + h * 128 + q];
  }
  const float bias0 = a.bias[lane & 31], bias1 = a.bias[32 + (lane & 31)], biasd = a.bias[64];
  __syncthreads();

  const long n_mt = (npix + 31) / 32;
  const int ics = a.in_cstride, ocs = a.out_cstride;
  for (long mt = (long)blockIdx.x * 4 + wave; mt < n_mt; mt += (long)gridDim.x * 4) {
    const long p = mt * 32 + (lane & 31);
    const bool valid = p < npix;
    const float* row = a.in + (size_t)(valid ? p : 0) * ics + a.in_coff;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = bias0; acc1[r] = bias1; }
    float d = biasd;
    f32x4 xa[C1_RING][2];      // ring: input channels 8 g .. 8 g + 7 of this lane's pixel
#pragma unroll
    for (int g = 0; g < C1_RING; ++g) {
      xa[g][0] = *reinterpret_cast<const f32x4*>(row + g * 8);
      xa[g][1] = *reinterpret_cast<const f32x4*>(row + g * 8 + 4);
    }
#pragma unroll 1
    for (int g0 = 0; g0 < C1_C8; g0 += C1_RING) {
#pragma unroll
      for (int gg = 0; gg < C1_RING; ++gg) {
        const int g = g0 + gg;
        const f32x4 x0 = xa[gg][0], x1 = xa[gg][1];
        if (g + C1_RING < C1_C8) {      // the slot is free: request the group one ring ahead
          xa[gg][0] = *reinterpret_cast<const f32x4*>(row + (g + C1_RING) * 8);
          xa[gg][1] = *reinterpret_cast<const f32x4*>(row + (g + C1_RING) * 8 + 4);
        }
        const f32x4 b0 = reinterpret_cast<const f32x4*>(wsm)[(0 * C1_C8 + g) * 64 + lane];
        const f32x4 b1 = reinterpret_cast<const f32x4*>(wsm)[(1 * C1_C8 + g) * 64 + lane];
        // k-step q of the group multiplies channel 8 g + 2 q + hh
        const float a0 = hh ? x0[1] : x0[0], a1 = hh ? x0[3] : x0[2], a2 = hh ? x1[1] : x1[0], a3 = hh ? x1[3] : x1[2];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1[0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b0[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b1[2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b0[3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b1[3], acc1, 0, 0, 0);
        // the dustbin: channels 8 g .. 8 g + 7 in ascending order (both lane halves hold the pixel's whole group)
        const f32x4 we = reinterpret_cast<const f32x4*>(wdl)[g * 2], wo = reinterpret_cast<const f32x4*>(wdl)[g * 2 + 1];
        d = __builtin_fmaf(x0[0], we[0], d);   d = __builtin_fmaf(x0[1], wo[0], d);
        d = __builtin_fmaf(x0[2], we[1], d);   d = __builtin_fmaf(x0[3], wo[1], d);
        d = __builtin_fmaf(x1[0], we[2], d);   d = __builtin_fmaf(x1[1], wo[2], d);
        d = __builtin_fmaf(x1[2], we[3], d);   d = __builtin_fmaf(x1[3], wo[3], d);
      }
    }
    // C layout: column = lane & 31 (channel), row i = (r & 3) + 8 (r >> 2) + 4 hh (pixel mt * 32 + i)
    float* out = a.out + a.out_coff;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long pp = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (pp < npix) {
        out[(size_t)pp * ocs + (lane & 31)] = acc0[r];
        out[(size_t)pp * ocs + 32 + (lane & 31)] = acc1[r];
      }
    }
    if (valid && hh == 0) out[(size_t)p * ocs + 64] = d;
  }
}

// convPb through the kernel above when the tensors are flat pixel lists (no padding between images); otherwise hipErrorNotSupported
hipError_t launch_conv1x1_256_65(const ConvArgs& a, hipStream_t s) {
  const long hw = (long)a.H * a.W;
  if (a.cout_real != 65 || a.in_img_stride != hw * a.in_cstride || a.out_img_stride != hw * a.out_cstride || a.out_cstride < 65 ||
      (a.in_cstride & 3) || (a.in_coff & 3) || a.in_coff + 256 > a.in_cstride)
    return hipErrorNotSupported;
  const int ncu = a.ncu > 0 ? a.ncu : 256;     // ConvArgs::ncu: the handle's device (no process-wide cache: one process may drive several GPUs)
  const long npix = hw * a.n_img;
  const long n_wg = (npix + 127) / 128;
  const int grid = (int)(n_wg < 2 * ncu ? n_wg : 2 * ncu);
  constexpr size_t lds = (size_t)(2 * C1_C8 * 256 + C1_C8 * 8) * sizeof(float);      // 65 KiB: two workgroups per CU
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_256_65_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(conv1x1_256_65_kernel, dim3(grid), dim3(256), lds, s, a, npix);
  return hipGetLastError();
}

}  // namespace d2fe
