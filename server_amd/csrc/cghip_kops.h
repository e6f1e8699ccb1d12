/*
 * cghip_kops.h -- the per-frame "program" the fused compositor kernel executes.
 *
 * The host planner (cghip_plan.cpp) turns each reference draw call (image_kernel::draw,
 * accelerator/ogl/image/image_kernel.cpp:134-420) into one KOp; the kernel (cghip_kernels.hip) walks the
 * KOp array once per output pixel and keeps every render surface of the reference renderer
 * (target / layer / mix / key textures, image_mixer.cpp:95,128,177,187) in registers.
 * POD only: the array is copied to device memory and read through scalar loads (all fields are
 * wave-uniform).
 */
#ifndef CGHIP_KOPS_H
#define CGHIP_KOPS_H

#include <stdint.h>

enum KOpKind : int32_t {
    KOP_CLEAR        = 0, /* surface := 0 (glClearTexImage of a fresh temporary) */
    KOP_DRAW_PLANES  = 1, /* item draw: sample source planes through the quad */
    KOP_DRAW_SURFACE = 2, /* full-frame 1:1 draw of a temporary surface (image_mixer.cpp:207-225) */
    /* Planner-internal markers, never seen by a kernel (cghip_plan.cpp removes them when it cuts the frame into
       launches). The kernels hold two key surfaces; a frame with more key textures alive at once (nested sublayers with
       keys of their own: image_mixer.cpp:104-114,153) parks one in HBM: KOP_SAVE_KEY says that the key texture
       `src_surface` (a surface SLOT of the draw list) leaves register `dst` here - the launch in front of the marker
       stores it -, KOP_LOAD_KEY that it returns into register `dst` - the launch behind the marker loads it. */
    KOP_SAVE_KEY = 3,
    KOP_LOAD_KEY = 4
};

enum KOpFlags : uint32_t {
    KF_LEVELS      = 1u << 0,
    KF_CSB         = 1u << 1,
    KF_CHROMA      = 1u << 2,
    KF_CHROMA_MASK = 1u << 3,
    KF_INVERT      = 1u << 4,
    KF_IS_HD       = 1u << 5,
    KF_PROJECTIVE  = 1u << 6, /* texture_q != 1 somewhere: divide s,t by q per pixel */
    KF_RECT        = 1u << 7, /* axis-aligned quad with one affine map: coverage == bbox, no edge tests */
    KF_UNIT_SCALE  = 1u << 8, /* KF_RECT and plane 0 maps 1:1 onto output pixels at an integer offset */
    KF_ADDITIVE    = 1u << 9, /* keyer::additive */
    KF_EDGE        = 1u << 11, /* fast kernel: x0 or x1 is not a multiple of 4, lanes on the box edge mask their pixels */
    KF_SOLID       = 1u << 10, /* KF_RECT and the source is one 1x1 packed texel (colour producer): every sample is it */
    KF_AXIS        = 1u << 13, /* KF_RECT, not projective, and s depends on x alone, t on y alone (exact zeros) */
    KF_RED_ONLY    = 1u << 12  /* fast kernel: planar draw into a key surface without colour adjustments: only red and
                                  alpha are looked at (image_mixer.cpp:174-183), the Cb plane is not read */
};

enum KFastClass : int32_t {
    FC_NONE   = 0,
    FC_PACKED = 1, /* one 4-channel plane (bgra / rgba / argb / abgr), unit scale */
    FC_PLANAR = 2, /* ycbcr / ycbcra planes of one bit depth, chroma at 1/1 or 1/2 per axis, unit scale */
    FC_SURFACE = 3, /* KOP_DRAW_SURFACE: a layer / mix temporary composited 1:1 (any blend mode, either keyer) */
    FC_BYTES   = 6, /* gray, luma, bgr, rgb: one 8-bit plane of 1 or 3 channels at unit scale */
    FC_UYVY    = 5, /* 8-bit uyvy at unit scale: one buffer bound as RG8 w x h and RGBA8 (w/2) x h */
    FC_SAMPLED = 4  /* planes in one of the straight-line sampling layouts (KSampleClass), axis-aligned box at
                       any scale or 1:1 off the 4-texel grid: four bilinear fetches per lane, coverage per pixel */
};

/* Which straight-line bodies of the 4-pixels-per-lane kernel run a draw (host: classify_fast). The kernel switches on
 * these two numbers and nothing else: computing them per op and wave from the flags cost ~40 scalar instructions. */
enum KFetchBody : int32_t {
    FB_SURFACE  = 0, /* KOP_DRAW_SURFACE: the source is a register surface */
    FB_PACKED8  = 1,
    FB_PACKED16 = 2,
    FB_PLANAR8  = 3, /* + csx + 2 * csy */
    FB_PLANAR16 = 7, /* + csx + 2 * csy */
    FB_UYVY8    = 11, /* FC_UYVY */
    FB_RECT16   = 16, /* the same four for 16-bit planes (fp32 filter weights) */
    FB_RECT_UYVY8 = 20, /* and for 8-bit uyvy (SC_UYVY8): + 1 if KF_UNIT_SCALE */
    FB_SAMPLED_END = 22, /* [FB_RECT8, FB_SAMPLED_END): the sampled bodies, compiled into the kernels' _s / _x instances */
    FB_BYTES8   = 24, /* FC_BYTES: gray / luma (1 channel) and bgr / rgb (3 channels), 8-bit, unit scale */
    FB_RECT8    = 12  /* FC_SAMPLED with KF_AXIS or KF_UNIT_SCALE and 8-bit planes: + 0 ycbcr(a) planes, + 1 one packed
                         plane, + 2 if KF_UNIT_SCALE (a 1:1 source off the aligned bodies' 4-texel grid) */
};
enum KBlendBody : int32_t {
    BB_RGBA_SIMPLE  = 0, /* no key, no invert, linear keyer */
    BB_RGBA_ONE_KEY = 1, /* exactly one key, else like 0 */
    BB_RGBA_GENERAL = 2,
    BB_KEY_SIMPLE   = 3, /* into a key surface */
    BB_KEY_GENERAL  = 4,
    BB_MODE         = 5, /* a temporary composited with a blend mode */
    BB_FUSED        = 6, /* KOp::fused: the item over zero, its UNORM store, the composite's blend mode */
    BB_MIX_HEAD     = 7  /* the first item of a fused MIX pair (cghip_plan.cpp: fuse_mix_pairs): this op, the next (the second
                            item) and the one after (the composite of the mix texture) are ONE step of the kernel, the mix
                            texture lives in registers of that step only. Launches with such ops run
                            cghip_composite_fast_p (KSpill::mix_pairs) */
};

/* How cghip_composite samples a KOP_DRAW_PLANES source (host classification, cghip_plan.cpp). The two common
 * layouts have straight-line bilinear bodies; everything else takes the generic per-format path. */
enum KSampleClass : int32_t {
    SC_GENERIC  = 0,
    SC_PLANAR8  = 1, /* ycbcr / ycbcra, every plane 1 channel 8 bit, Cb and Cr of one size, alpha the size of Y */
    SC_PACKED8  = 2, /* bgra / rgba / argb / abgr in one 4-channel 8-bit plane */
    SC_PLANAR16 = 3, /* the same two layouts with 16-bit samples (fp32 filter weights) */
    SC_PACKED16 = 4,
    SC_UYVY8    = 5  /* 8-bit uyvy: luma from the RG8 view (.g), Cb / Cr from the RGBA8 view of half the width (.b / .r) */
};

/* register-resident surfaces the kernel has names for */
enum KSurf : int32_t {
    KS_NONE = -1,
    KS_RGBA0 = 0, /* frame target */
    KS_RGBA1 = 1,
    KS_RGBA2 = 2,
    KS_KEY0  = 3, /* 1-channel */
    KS_KEY1  = 4,
    KS_COUNT = 5
};

/* A frame whose draws fall into different kernel classes is composited by several launches, one per run of draws
 * of one class (cghip_plan.cpp: segments). The register surfaces that are live across a launch boundary travel
 * through HBM in the frame's own format: RGBA0 in the frame buffer itself, the others in these buffers. Stored values
 * are UNORM codes, which is exactly what the registers hold (every draw ends in a UNORM store), so the hand-over is
 * lossless. Bit s of the masks = KSurf s. gx0, gy0: pixel origin of the launch grid (a launch may cover only the box
 * its draws can touch). */
struct KSpill {
    void*    rgba1;
    void*    rgba2;
    void*    key0; /* what KEY0 / KEY1 are LOADED from ... */
    void*    key1;
    void*    key0_st; /* ... and STORED to: key textures parked in HBM keep their buffer while the register moves on */
    void*    key1_st;
    uint32_t load_mask;  /* surfaces to read before the first op */
    uint32_t store_mask; /* surfaces to write after the last op */
    int32_t  gx0, gy0;
    int32_t  nbx, nby;   /* workgroups per row of the launch and rows of workgroups (set by the launcher, see xcd_block) */
    /* Which ops of this launch can touch which part of the frame (host: cghip_plan.cpp, op_masks): bit i of
       masks[chunk * (ncols + nrows) + col] & masks[chunk * (ncols + nrows) + ncols + row] is set iff op 64 * chunk + i
       has pixels in column band `col` (x >> col_shift) and in row band `row` (y >> 2). A wave walks only those ops. */
    const uint64_t* masks;
    int32_t  ncols, nrows, nchunks, col_shift;
    int32_t  dense; /* every op is in every band (full-screen stacks): the table is not read */
    int32_t  one_temp; /* touches RGBA1 but not RGBA2 (loads and stores included): the fast kernel with one temporary;
                          2: and no op with a blend mode (a plain MIX transition): the instance without the blend modes */
    int32_t  fused; /* the launch holds a fused blend-mode layer (KOp::fused): picks the general kernel's instance */
    int32_t  mono;  /* lean 4-pixels-per-lane launches: the KFetchBody every draw of the launch takes, or -1 */
    int32_t  mix_pairs; /* the launch holds BB_MIX_HEAD ops: the lean kernel's instance that knows them */
    int32_t  yuv_lut;  /* 8-bit build: the launch has aligned draws of 8-bit planar / uyvy sources, whose YCbCr terms come from
                          tables every workgroup builds in LDS first (cghip_kernels.hip: YuvLut); 0: no tables, no barrier */
    int32_t  host_out; /* the frame buffer is pinned host memory (direct readback): plain stores, which the L2 merges
                          into whole lines before they cross PCIe; frames that stay in HBM leave through non-temporal
                          stores (written once, read by nobody on the GPU) */
};

struct KPlane {
    const void* ptr;
    int32_t     w, h;
    int32_t     channels; /* 1..4 samples per texel */
    int32_t     bits;     /* 8 or 16 */
    float       fw, fh;   /* (float)w, (float)h */
};

/* One triangle of the quad (image_kernel.cpp:383-384) in pixel-index space: pixel (x,y) is covered iff
 * all three e[k][0]*x + e[k][1]*y + e[k][2] >= 0 (fixed-point edge functions, fill-rule bias folded
 * into the constant; exact in double). Varyings: v = a[j][0] + a[j][1]*x + a[j][2]*y for
 * j = s*q, t*q, q. */
struct KTri {
    double e[3][3];
    double a[3][3];
};

struct KOp {
    int32_t  kind;
    uint32_t flags;
    int32_t  dst;         /* KSurf */
    int32_t  src_surface; /* KSurf for KOP_DRAW_SURFACE */
    int32_t  local_key;   /* KSurf or KS_NONE */
    int32_t  layer_key;
    int32_t  pixel_format;
    int32_t  blend_mode;
    int32_t  nplanes;
    int32_t  ntri;
    /* pixels that can be touched: coverage bbox ∩ scissor ∩ frame; [x0,x1) x [y0,y1) */
    int32_t x0, y0, x1, y1;
    int32_t unit_dx, unit_dy; /* KF_UNIT_SCALE: source texel = (x + unit_dx, y + unit_dy) */
    float   opacity;
    float   lv_min_input, lv_inv_range, lv_inv_gamma, lv_min_output, lv_max_output;
    float   brt, sat, con;
    float   ch_target_hue, ch_hue_width, ch_min_saturation, ch_min_brightness, ch_softness, ch_spill,
        ch_spill_saturation;
    /* fast-path classification (host, cghip_plan.cpp): 0 = general kernel only; FC_* otherwise */
    int32_t fast_class;
    float   ch_inv_spill, ch_inv_soft; /* 1/ch_spill, 1/(ch_softness - 1) */
    int32_t sample_class; /* KSampleClass */
    int32_t csx, csy; /* FC_PLANAR: chroma planes are 1/2 width (csx) and/or 1/2 height (csy) of plane 0 */
    /* divisors of the shader's per-pixel divisions by uniforms, next to their reciprocals above: x / c is evaluated as
       div_const(x, c, 1/c), correctly rounded; x * (1/c) alone is an ulp off in a third of the cases and, with round
       numbers like levels 0.1 / 0.9, moved 1 % of 8-bit samples (tests/test_fuzz.py) */
    float lv_range;      /* max_input - min_input */
    float ch_soft_range; /* ch_softness - 1 */
    /* FC_PLANAR with 16-bit planes, for the kernels of 16-bit frames: sample codes -> colour as one affine map
       (cghip_kernels.hip: fetch_planar16r): aY, aR, cR, aGr, aGb, cG, aB, cB */
    float ycoef[8];
    int32_t fetch_body, blend_body; /* KFetchBody, KBlendBody (fast kernel) */
    /* A blend-mode layer that holds ONE item is three reference draws: the layer texture is cleared, the item drawn
       into it, and the texture composited over the target with the layer's mode (image_mixer.cpp:128-171,207-225).
       The planner fuses them into this one op (cghip_plan.cpp: fuse_layers): the item's colour
       goes through the UNORM store the texture would have given it, then through the composite's blend, and the
       texture never exists. x0..y1 is then the COMPOSITE's box, ax0..ay1 the item's own; b_* are the composite's
       keys, opacity, keyer / invert flags and mode. Unfused ops: fused = 0, ax0..ay1 = x0..y1. */
    int32_t  fused, b_blend_mode, b_local_key, b_layer_key;
    uint32_t b_flags;
    float    b_opacity;
    int32_t  ax0, ay0, ax1, ay1;
    KPlane  planes[4];
    KTri    tri[2];
};

#endif
