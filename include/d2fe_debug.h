/*
 * d2fe_debug.h -- test hooks and kernel diagnostics of the DEVELOPMENT library (d2slam_amd/lib/libd2fe_hip_dev.so: the same sources as
 * libd2fe_hip.so compiled with -DD2FE_DEVTOOLS; `python -m d2slam_amd.build --dev`).  The product library exports none of these symbols,
 * reads no environment variable and contains no ablation or trace code.  The development library additionally honours the D2FE_*
 * environment switches listed in tools/README.md (alternate kernel schedules for A/B measurements, phase stamps).
 */
#ifndef D2FE_DEBUG_H_
#define D2FE_DEBUG_H_

#include "d2fe.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug/inspection: copy an internal device tensor of the last extract call to the host.
 * names: "conv1a".."conv4b","convPaDa","logits","desc_raw","semi".  Returns bytes copied or <0. */
D2FE_API long d2fe_debug_read(d2fe_handle h, const char* name, void* dst, size_t max_bytes);
/* NetVLAD inspection: output of layer `layer` (NHWC fp32, n_images of the handle's maximum size) of the last d2fe_netvlad* call, when the
 * execution plan materialises it -- the last layer of every fused MobileNetV2 block and every unfused layer; D2FE_ERR_NOT_READY for a
 * layer that only exists in LDS inside a fused block (D2FE_NV_LEGACY=1 in the environment of the development library: one launch per layer, everything readable). */
D2FE_API long d2fe_debug_netvlad_layer(d2fe_handle h, int layer, int n_images, void* dst, size_t max_bytes);
/* Kernel diagnostics (tools/nv_stamps.py): with D2FE_NV_STAMP_STEP=<execution-plan step> in the environment when the network was loaded, the
 * wall_clock64() phase stamps [workgroup][32] (100 MHz ticks) that step's block kernel wrote during the last d2fe_netvlad* call.  Returns the number
 * of workgroups copied (<= max_wgs) or <0. */
D2FE_API long d2fe_debug_netvlad_stamps(d2fe_handle h, unsigned long long* dst, long max_wgs);
/* One 3x3 / pad 1 layer (cin 64 or 128, ReLU, optional 2x2 max-pool) through the Winograd kernels of D2FE_PREC_F32_WINO, host
 * NHWC buffers in and out; iters > 0 also times `iters` back-to-back launches (HIP events on the handle's stream).  For the
 * layer-level parity tests (tests/test_wino.py) and tools/; the product path is d2fe_superpoint_extract*. */
/* The host-side weight transform of that mode (U = G g G^T, packed [32-channel group][k-step][row i][lane][4]); needs no GPU.
 * Returns the number of floats written (16 * cin * cout rounded up to 64 channels). */
/* Tile shape the NetVLAD block launchers pick for an Ho x Wo output map (needs no GPU): kind 0 stride-1 blocks, 1 the first block (stride = the first
 * conv's), 2 nv_xblock_kernel (stride 1 or 2). */
D2FE_API int d2fe_debug_netvlad_tile(int kind, int Ho, int Wo, int stride, int* th, int* tw);
/* The host-side weight packing of the NetVLAD block kernels (needs no GPU; tests/test_netvlad_pack_cpu.py): kind 0..5 = expand / depthwise + project
 * records of nv_pblock_kernel, nv_xblock_kernel, nv_tail_kernel (see csrc/api.hip).  Returns the number of floats written or <0. */
D2FE_API long d2fe_debug_pack_netvlad(int kind, const float* we, const float* be, const float* wd, const float* bd, const float* wp, int cin, int chid,
                                      int cout, float* out, long max_floats);
D2FE_API long d2fe_debug_pack_wino(const float* weight /*[cout][cin][3][3]*/, int cout, int cin, float* out, long max_floats);
D2FE_API int d2fe_debug_conv3x3_wino(d2fe_handle h, const float* in, int n, int H, int W, int cin, const float* weight,
                                     const float* bias, int cout, int pool, int relu, float* out, int iters, float* ms_per_launch);

/* Host-pointer calls replay cached hipGraphs of their launch sequences from the third call with the same geometry on (D2FE_GRAPH=0 in the
 * environment turns that off).  Returns how many graphs the handle holds; *rejected (may be NULL) = geometries whose capture failed and which
 * therefore keep launching kernel by kernel (diagnostic). */
D2FE_API int d2fe_debug_graph_count(d2fe_handle h, int* rejected);

/* The wall_clock64() phase stamps [workgroup][16] of the last d2fe_match_batch_device launch made with D2FE_MATCH_STAMPS=1 in the environment
 * (tools/match_stamps.py).  Returns the number of workgroups copied or <0. */
D2FE_API long d2fe_debug_match_stamps(d2fe_handle h, unsigned long long* dst, long max_wgs);

#ifdef __cplusplus
}
#endif
#endif /* D2FE_DEBUG_H_ */
