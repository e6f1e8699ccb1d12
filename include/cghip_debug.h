/*
 * cghip_debug.h -- introspection entry points of libcghip.so used by the test-suite only. They have no
 * counterpart in the reference (which has no tests) and no production caller.
 */
#ifndef CGHIP_DEBUG_H
#define CGHIP_DEBUG_H

#include "cghip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Runs the host planner (image_kernel::draw host math + register binding) and copies out the kernel
 * program; needs no GPU. kops_out receives records of cghip_debug_kop_size() bytes each. */
int cghip_debug_plan(const cghip_format* fmt, const cghip_surface* surfaces, int nsurfaces, const cghip_draw* draws,
                     int ndraws, void* kops_out, int capacity, int* nops, int* full);
int cghip_debug_kop_size(void);
/* In *full above: bit 0 = the frame uses RGBA temporaries, bit 1 = every draw is of the 4-pixels-per-lane class,
 * bits 8.. = number of kernel launches (segments) the frame is cut into.
 * The launches of a planned frame, 14 ints each: first op, op count, fast, full, load mask, store mask, the pixel
 * box gx0, gy0, gx1, gy1 the launch covers, and the key buffers KEY0 / KEY1 are loaded from and stored to (-1: none).
 * Returns the number of launches (or an error < 0). */
int cghip_debug_plan_segments(const cghip_format* fmt, const cghip_surface* surfaces, int nsurfaces,
                              const cghip_draw* draws, int ndraws, int* out, int capacity);
/* At most n launches per frame (1 = never cut a frame; default 8). */
void cghip_debug_max_segments(int n);
/* Scales the planner's estimate of what a cut costs (default 1.0). 0 cuts wherever the kernel class changes, which
 * is how the tests get small frames cut at all. */
void cghip_debug_cut_cost_scale(double scale);
/* A plane descriptor with no device memory behind it, for planner tests. Release with cghip_plane_release. */
int cghip_debug_plane_stub(int width, int height, int channels, int bits, cghip_plane** out);

/* Plans every frame as one launch of the general one-pixel-per-lane kernel, so tests can compare it with the
 * 4-pixels-per-lane kernel and with frames cut into several launches. */
void cghip_debug_force_general_kernel(int on);
/* The planner's KOP_SAVE_KEY / KOP_LOAD_KEY markers (a key texture leaves / re-enters a key register: frames with more
 * than two key textures alive at once) stay in the op list cghip_debug_plan returns, so tests can replay them. */
void cghip_debug_keep_pseudo_ops(int on);
/* The planner shrinks the composite of a layer / mix temporary to the box that was drawn into the temporary (outside it
 * the draw is the identity); on = composites cover the whole frame as in the reference, so tests can compare. */
void cghip_debug_full_frame_composites(int on);
/* The planner drops draws that a later full-frame opaque layer hides (and keys / temporaries only they consumed);
 * on = every draw of the reference is executed, so tests can compare. */
void cghip_debug_keep_hidden_ops(int on);
/* The general kernel samples the common plane layouts (KSampleClass) with straight-line bodies; on = every draw takes
 * the generic per-format path instead, so tests can compare the two. */
void cghip_debug_generic_sampling(int on);
/* Launches the 4-pixels-per-lane kernel with this much unused dynamic LDS, which caps the workgroups resident on a
 * CU (160 KB each): occupancy experiments only (tests/gpu_occupancy.py). 0 = off. */
void cghip_debug_pad_lds(int bytes);
/* Rows (= waves) per workgroup of the 4-pixels-per-lane kernel, 1..4 (default 4): launch-shape experiments. */
void cghip_debug_fast_rows(int rows);
/* Large launches of the 4-pixels-per-lane kernel number their tiles so that neighbouring tiles share an XCD's L2
 * (cghip_kernels.hip: xcd_block); on = tiles in dispatch order as in small launches, to measure the difference. */
void cghip_debug_plain_tiles(int on);
/* Launches whose draws all take one fetch body run a kernel instance compiled for that body alone
 * (cghip_composite_fast_m); on = they run the kernel with all bodies, so tests can compare the two bit for bit. */
void cghip_debug_no_mono_kernels(int on);
/* A MIX transition between aligned unit-scale sources (CLEAR, two items into the mix texture, its composite) is planned
 * as one step of the lean 4-pixels-per-lane kernel when the frame needs no temporary otherwise (cghip_plan.cpp:
 * fuse_mix_pairs); on = it stays four ops in the kernels that hold a temporary, so tests can compare bit for bit. */
void cghip_debug_no_mix_fusion(int on);

#ifdef __cplusplus
}
#endif
#endif
