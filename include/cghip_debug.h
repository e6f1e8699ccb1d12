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
/* A plane descriptor with no device memory behind it, for planner tests. Release with cghip_plane_release. */
int cghip_debug_plane_stub(int width, int height, int channels, int bits, cghip_plane** out);

/* Forces the general one-pixel-per-lane kernel even for frames the 4-pixels-per-lane kernel could run, so
 * tests can compare the two. */
void cghip_debug_force_general_kernel(int on);
/* Launches the 4-pixels-per-lane kernel with this much unused dynamic LDS, which caps the workgroups resident on a
 * CU (160 KB each): occupancy experiments only (tests/gpu_occupancy.py). 0 = off. */
void cghip_debug_pad_lds(int bytes);
/* Rows (= waves) per workgroup of the 4-pixels-per-lane kernel, 1..4 (default 4): launch-shape experiments. */
void cghip_debug_fast_rows(int rows);

#ifdef __cplusplus
}
#endif
#endif
