/*
 * cghip_kernels.hip -- the fused gfx950 compositor: ONE launch per frame replaces every
 * glDrawArrays of the reference renderer (accelerator/ogl/image/image_kernel.cpp:402) and the
 * fragment shader they run (accelerator/ogl/image/shader.frag:523-542).
 *
 * Mapping (wave64): one wave owns 64*P consecutive pixels of ONE output row, a 256-thread workgroup owns
 * 4 consecutive rows of the same column span, so source rows shared by vertically adjacent outputs
 * (bilinear taps, 4:2:0 chroma) are served by the CU's L1. Every lane walks the frame's KOp list
 * (wave-uniform, scalar loads) and keeps all render surfaces of the reference (target, layer, mix and
 * key textures) in VGPRs; HBM traffic is source texels in, one BGRA pixel out.
 *
 * Numerics follow what the reference's shaders do under a GL implementation (goldens: Mesa llvmpipe):
 *  - the store after EVERY draw rounds to UNORM8/16 (round-to-nearest-even of x*(2^n-1)), emulated in
 *    registers between ops, because every reference draw lands in an RGBA8 texture;
 *  - 8-bit textures are filtered with 8-bit weights and rounded integer lerps, 16-bit ones in fp32;
 *  - coverage uses 24.8 fixed-point edge functions with the top-left rule (exact, in double);
 *  - this file is compiled with -ffp-contract=off (csrc/Makefile): llvmpipe's JIT never fuses a multiply with
 *    an add, and with contraction on about 2e-4 of 8-bit samples came out 1 LSB off, which a steep blend
 *    (overlay, dodge) can double. The only fmas are explicit __builtin_fmaf calls whose single rounding is
 *    proven equal to the reference's sequence (div_const, fdiv, quantize). With it every 8-bit golden scene
 *    is bit-exact against the oracle; it costs ~5% of the kernel time.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

/* This file is compiled TWICE (csrc/Makefile): with -DCGHIP_FRAME_BITS=8 -ffp-contract=off into the kernels of 8-bit
 * frames (bit-exact against the reference pipeline), and with -DCGHIP_FRAME_BITS=16 -ffp-contract=fast into the
 * kernels of 16-bit frames (relaxed numerics, see divf below). Each object carries its own launcher
 * (cghip_launch_composite8 / 16); the 8-bit object also holds what exists once: the dispatcher, the debug switches
 * and the scratch array of the allocation-free entry point. */
#ifndef CGHIP_FRAME_BITS
#error "compile with -DCGHIP_FRAME_BITS=8 or 16 (see csrc/Makefile)"
#endif

#include "../../include/cghip.h" /* CGHIP_NOALLOC_SCRATCH_BYTES */
#include "cghip_kops.h"

#include <mutex>
#include <type_traits>

#define DEV __device__ __forceinline__

/* relaxed numerics: the object for 16-bit frames (see divf below) */
constexpr bool RXF = CGHIP_FRAME_BITS == 16;
DEV int opaque_s(int v); /* below: a scalar the compiler may not look through */
/* A per-pixel block of a 4-pixel body behind a branch of its own (always taken; the compiler cannot know). Measured,
 * not understood: the chroma key of four pixels as ONE straight-line block of 258 instructions ran slower than the same
 * code as four blocks with a scalar branch in front of each (16 chroma-key layers at 2160p, 16-bit: 0.295 -> 0.279 ms;
 * 8-bit 0.440 -> 0.395), although tests/ubench finds no cost in dependent chains or register banks at 7 waves. */
#define PX_BRANCH if (opaque_s(1))

/* compile-time loop: the per-pixel register arrays must never be indexed dynamically (that would put them
 * in scratch memory) */
template <int I, int N, class F>
DEV void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct f4 {
    float x, y, z, w;
};
struct f3 {
    float x, y, z;
};

/* ------------------------------------------------------------------ UNORM store emulation */

template <int BITS>
struct Q {
    static constexpr float scale   = BITS == 8 ? 255.0f / 256.0f : 65535.0f / 65536.0f;
    static constexpr float magic   = BITS == 8 ? 32768.0f : 128.0f;
    static constexpr float dequant = BITS == 8 ? 256.0f / 255.0f : 65536.0f / 65535.0f; /* == 2^n * fl(1/(2^n-1)) */
    static constexpr float unit    = BITS == 8 ? 256.0f : 65536.0f;
};

/* float -> the float a later texture fetch of the stored UNORM would return: clamp, scale, add the
 * magic constant (leaves exactly n fraction bits, round-to-nearest-even), subtract it again. */
template <int BITS>
DEV float quantize(float x)
{
    x             = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); /* NaN -> 0 like the clamp before a UNORM store */
    const float r = x * Q<BITS>::scale + Q<BITS>::magic; /* two roundings, never an fma (pragma above) */
    /* t = r - magic = k / 2^n exactly, result = fl(t * dequant) = fl(k * fl(1/(2^n-1))). One explicit fma does
       both steps with the same single rounding: r*dequant - magic*dequant is (r - magic)*dequant exactly, and
       magic*dequant is a power-of-two multiple of dequant, hence representable. */
    return __builtin_fmaf(r, Q<BITS>::dequant, -Q<BITS>::magic * Q<BITS>::dequant);
}
template <int BITS>
DEV uint32_t to_code(float q)
{ /* inverse of the dequant step: q = k*fl(1/(2^n-1)) -> k */
    return (uint32_t)__builtin_rintf(q * (BITS == 8 ? 255.0f : 65535.0f));
}

/* ------------------------------------------------------------------ texel fetch + filtering */

DEV int imin(int a, int b) { return a < b ? a : b; }
DEV int imax(int a, int b) { return a > b ? a : b; }
DEV int iclamp(int v, int lo, int hi) { return imin(imax(v, lo), hi); }

/* plane memory is hipMalloc'd global memory: say so, or the loads through pointers fetched from the op list
 * compile to flat_load (generic address space), which also ties up the LDS/scalar wait counter */
#define GLOBAL_AS __attribute__((address_space(1)))
typedef const GLOBAL_AS uint8_t*  G8;
typedef const GLOBAL_AS uint16_t* G16;
typedef const GLOBAL_AS uint32_t* G32;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const GLOBAL_AS u32x2*    GU2;
typedef const GLOBAL_AS u32x4*    GU4;
typedef const GLOBAL_AS char*     GB;

/* load through a wave-uniform base plus a 32-bit per-lane byte offset: the SGPR-base + VGPR-offset form of
 * global_load, instead of a 64-bit multiply-add per lane and load */
template <class T>
DEV T load_at(GB base, uint32_t byte_off)
{
    return *(const GLOBAL_AS T*)(base + byte_off);
}

/* 8-bit texel as a packed word in MEMORY channel order (byte k = k-th sample of the texel) */
DEV uint32_t load8(const KPlane& p, int x, int y)
{
    const size_t   idx = (size_t)y * (size_t)p.w + (size_t)x;
    const G8       b   = (G8)p.ptr;
    switch (p.channels) {
        case 1: return b[idx];
        case 2: return ((G16)p.ptr)[idx];
        case 3: return (uint32_t)b[idx * 3] | ((uint32_t)b[idx * 3 + 1] << 8) | ((uint32_t)b[idx * 3 + 2] << 16);
        default: return ((G32)p.ptr)[idx];
    }
}

/* a + ((w*(b-a) + 128) >> 8) on four packed bytes at once: (a*(256-w) + b*w + 128) >> 8 per byte */
DEV uint32_t lerp8x4(uint32_t a, uint32_t b, uint32_t w)
{
    const uint32_t iw = 256u - w;
    uint32_t       e  = ((a & 0x00FF00FFu) * iw + (b & 0x00FF00FFu) * w + 0x00800080u) >> 8;
    uint32_t       o  = ((a >> 8) & 0x00FF00FFu) * iw + ((b >> 8) & 0x00FF00FFu) * w + 0x00800080u;
    return (e & 0x00FF00FFu) | (o & 0xFF00FF00u);
}

/* memory-order samples -> GL texel (r,g,b,a): strides 3/4 were uploaded as GL_BGR/GL_BGRA
 * (accelerator/ogl/util/texture.cpp:31-33) */
DEV f4 gl_texel(int channels, float m0, float m1, float m2, float m3)
{
    switch (channels) {
        case 1: return f4{m0, 0.0f, 0.0f, 1.0f};
        case 2: return f4{m0, m1, 0.0f, 1.0f};
        case 3: return f4{m2, m1, m0, 1.0f};
        default: return f4{m2, m1, m0, m3};
    }
}

/* GL_LINEAR + CLAMP_TO_EDGE fetch of an 8-bit plane at fixed-point texel coordinates (24.8, already
 * shifted by half a texel). */
DEV f4 sample8_fixed(const KPlane& p, int iu, int iv)
{
    int            x0 = iu >> 8, y0 = iv >> 8;
    const uint32_t wx = (uint32_t)(iu & 0xff), wy = (uint32_t)(iv & 0xff);
    const int      x1 = iclamp(x0 + 1, 0, p.w - 1), y1 = iclamp(y0 + 1, 0, p.h - 1);
    x0                = iclamp(x0, 0, p.w - 1);
    y0                = iclamp(y0, 0, p.h - 1);
    uint32_t a        = load8(p, x0, y0);
    if (wx != 0)
        a = lerp8x4(a, load8(p, x1, y0), wx);
    if (wy != 0) {
        uint32_t b = load8(p, x0, y1);
        if (wx != 0)
            b = lerp8x4(b, load8(p, x1, y1), wx);
        a = lerp8x4(a, b, wy);
    }
    const float sc = 1.0f / 255.0f;
    return gl_texel(p.channels, (float)(a & 0xff) * sc, (float)((a >> 8) & 0xff) * sc, (float)((a >> 16) & 0xff) * sc,
                    (float)(a >> 24) * sc);
}

DEV void load16(const KPlane& p, int x, int y, float m[4])
{
    const size_t    idx = ((size_t)y * (size_t)p.w + (size_t)x) * (size_t)p.channels;
    const G16       s   = (G16)p.ptr;
    const float     sc  = 1.0f / 65535.0f;
    m[0] = m[1] = m[2] = m[3] = 0.0f;
    switch (p.channels) {
        case 1: m[0] = (float)s[idx] * sc; break;
        case 2: {
            uint32_t v = *(G32)(s + idx);
            m[0] = (float)(v & 0xffff) * sc, m[1] = (float)(v >> 16) * sc;
            break;
        }
        case 3: m[0] = (float)s[idx] * sc, m[1] = (float)s[idx + 1] * sc, m[2] = (float)s[idx + 2] * sc; break;
        default: {
            const u32x2 v = *(GU2)(s + idx);
            m[0] = (float)(v.x & 0xffff) * sc, m[1] = (float)(v.x >> 16) * sc;
            m[2] = (float)(v.y & 0xffff) * sc, m[3] = (float)(v.y >> 16) * sc;
        }
    }
}

/* GL_LINEAR + CLAMP_TO_EDGE fetch of a 16-bit plane at float texel coordinates u,v (half-texel shift
 * applied): fp32 weights, v0 + w*(v1-v0), x first then y. */
DEV f4 sample16_float(const KPlane& p, float u, float v)
{
    const float fx = __builtin_floorf(u), fy = __builtin_floorf(v);
    const float wx = u - fx, wy = v - fy;
    int         x0 = (int)fx, y0 = (int)fy;
    const int   x1 = iclamp(x0 + 1, 0, p.w - 1), y1 = iclamp(y0 + 1, 0, p.h - 1);
    x0             = iclamp(x0, 0, p.w - 1);
    y0             = iclamp(y0, 0, p.h - 1);
    float a[4], b[4];
    load16(p, x0, y0, a);
    if (wx != 0.0f) {
        load16(p, x1, y0, b);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            a[k] = a[k] + wx * (b[k] - a[k]);
    }
    if (wy != 0.0f) {
        float c[4];
        load16(p, x0, y1, c);
        if (wx != 0.0f) {
            load16(p, x1, y1, b);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                c[k] = c[k] + wx * (b[k] - c[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            a[k] = a[k] + wy * (c[k] - a[k]);
    }
    return gl_texel(p.channels, a[0], a[1], a[2], a[3]);
}

/* texture(sampler, (s,t)) */
DEV f4 sample_norm(const KPlane& p, float s, float t)
{
    if (p.bits == 8) {
        const int iu = (int)__builtin_rintf(s * p.fw * 256.0f) - 128;
        const int iv = (int)__builtin_rintf(t * p.fh * 256.0f) - 128;
        return sample8_fixed(p, iu, iv);
    }
    return sample16_float(p, s * p.fw - 0.5f, t * p.fh - 0.5f);
}

/* KF_UNIT_SCALE: plane 0 maps 1:1 at an integer offset; plane k is w_k/w_0 of it. px,py = plane-0
 * texel index; rx = w_k/w_0, ry = h_k/h_0 (exact small dyadic ratios for 4:2:x video). */
DEV f4 sample_unit(const KPlane& p, int px, int py, float rx, float ry)
{
    if (p.bits == 8) {
        const int iu = (int)__builtin_rintf(((float)px + 0.5f) * (rx * 256.0f)) - 128;
        const int iv = (int)__builtin_rintf(((float)py + 0.5f) * (ry * 256.0f)) - 128;
        return sample8_fixed(p, iu, iv);
    }
    return sample16_float(p, ((float)px + 0.5f) * rx - 0.5f, ((float)py + 0.5f) * ry - 0.5f);
}

/* ------------------------------------------------------------------ shader.frag restated for gfx950 */

DEV float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
DEV float clamp01(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
DEV float fractf(float x) { return x - __builtin_floorf(x); }

/* x / c for a compile-time constant c, correctly rounded in 3 ops instead of the ~10 of a full divide:
 * q = x*rc, one Newton step on the residual (Markstein). */
DEV float div_const(float x, float c, float rc)
{
    const float q = x * rc;
    const float r = __builtin_fmaf(-q, c, x);
    return __builtin_fmaf(r, rc, q);
}

/* a / b, correctly rounded, for the finite, normal-range denominators this shader produces (alpha + 1e-7,
 * 6d + 1e-10, ...): v_rcp_f32 (1 ulp) is first refined to the correctly rounded reciprocal, then the quotient gets
 * one residual correction (Markstein). 6 issue slots + the rcp instead of the ~14 of the IEEE divide sequence, whose
 * extra work is scaling for denormal / huge operands that cannot occur here. A single Newton step on the raw rcp
 * was not enough: a random-scene test found a quotient one ulp off (1 LSB on a chroma-keyed gray pixel). */
DEV float fdiv(float a, float b)
{
    float       rb = __builtin_amdgcn_rcpf(b);
    const float e  = __builtin_fmaf(-b, rb, 1.0f);
    rb             = __builtin_fmaf(e, rb, rb);
    const float q  = a * rb;
    const float r  = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(r, rb, q);
}

/* ---- numerics of the 16-bit frame kernels (RX = true) ----------------------------------------------------------
 * 8-bit frames are bit-exact against the reference pipeline: every operation above and below follows the shader's
 * own sequence of roundings. 16-bit frames are the north star's extension with a tolerance of +-2 LSB16, and one
 * LSB16 is 2^-16: an fp32 rounding (2^-24) is 1/256 of it. There the kernels evaluate the same formulas with fewer
 * roundings instead of the same ones: multiply-adds are fused, x / c is x * (1 / c), a / b is a * rcp(b) (1 ulp),
 * the YCbCr matrix acts directly on the sample codes, and the UNORM store after every draw is three fmas (fused
 * blend with the clamp as an output modifier, round with the magic constant, back to [0,1]). The result differs from
 * the exact sequence by a few 1e-7, i.e. it rounds to the neighbouring code for a few per cent of the samples; the
 * draw-by-draw UNORM rounding itself is kept. tests/parity16.py bounds the MAXIMUM at +-2 LSB16 outside
 * the pixels where the scene itself amplifies such noise. Instruction count on the headline frame: 223 M -> see
 * DESIGN.md section 4. */
template <bool RX>
DEV float divf(float a, float b)
{
    if constexpr (RX)
        return a * __builtin_amdgcn_rcpf(b);
    else
        return fdiv(a, b);
}
template <bool RX>
DEV float divc(float x, float c, float rc)
{
    if constexpr (RX)
        return x * rc;
    else
        return div_const(x, c, rc);
}

/* shader.frag:441-467; internal order x=blue y=green z=red w=alpha */
DEV f4 ycbcra_to_rgba(float Y, float Cb, float Cr, float A, bool hd)
{
    const float yy = 1.164f * (Y * 255.0f - 16.0f);
    const float cb = Cb * 255.0f - 128.0f, cr = Cr * 255.0f - 128.0f;
    f4          c;
    if (hd) {
        c.z = divc<RXF>((yy + 1.793f * cr), 255.0f, 1.0f / 255.0f);
        c.y = divc<RXF>((yy - 0.534f * cr - 0.213f * cb), 255.0f, 1.0f / 255.0f);
        c.x = divc<RXF>((yy + 2.115f * cb), 255.0f, 1.0f / 255.0f);
    } else {
        c.z = divc<RXF>((yy + 1.596f * cr), 255.0f, 1.0f / 255.0f);
        c.y = divc<RXF>((yy - 0.813f * cr - 0.391f * cb), 255.0f, 1.0f / 255.0f);
        c.x = divc<RXF>((yy + 2.018f * cb), 255.0f, 1.0f / 255.0f);
    }
    c.w = A;
    return c;
}

/* shader.frag:304-321. step() yields exactly 0 or 1, and mix(a, b, 0|1) = a*(1-t) + b*t is then exactly a or b,
 * so the seven mixes are selects. */
template <bool RX = false>
DEV f3 rgb2hsv(f3 c)
{
    const bool  gb = !(c.y < c.z); /* step(c.b, c.g) == 1 */
    const f4    p  = {gb ? c.y : c.z, gb ? c.z : c.y, gb ? 0.0f : -1.0f, gb ? -1.0f / 3.0f : 2.0f / 3.0f};
    const bool  rp = !(c.x < p.x); /* step(p.x, c.r) == 1 */
    const f4    q  = {rp ? c.x : p.x, p.y, rp ? p.z : p.w, rp ? p.x : c.x};
    const float d  = q.x - __builtin_fminf(q.w, q.y);
    const float e  = 1.0e-10f;
    return f3{__builtin_fabsf(q.z + divf<RX>(q.w - q.y, 6.0f * d + e)), divf<RX>(d, q.x + e), q.x};
}
DEV f3 hsv2rgb(f3 c)
{
    const f3 p = {__builtin_fabsf(fractf(c.x + 1.0f) * 6.0f - 3.0f),
                  __builtin_fabsf(fractf(c.x + 2.0f / 3.0f) * 6.0f - 3.0f),
                  __builtin_fabsf(fractf(c.x + 1.0f / 3.0f) * 6.0f - 3.0f)};
    return f3{c.z * mixf(1.0f, clamp01(p.x - 1.0f), c.y), c.z * mixf(1.0f, clamp01(p.y - 1.0f), c.y),
              c.z * mixf(1.0f, clamp01(p.z - 1.0f), c.y)};
}

/* hue distance wrapped to [-0.5, 0.5] (shader.frag:349-353: +1 below -0.5, -1 above 0.5): the nearest integer, held to
 * -1..1, subtracted - three instructions for two compares, two adds and two selects, and the same number for every
 * input (the subtraction IS that add; rint(+-0.5) is 0 like the strict compares) */
DEV float wrap_hue(float diff)
{
    return diff - __builtin_amdgcn_fmed3f(__builtin_rintf(diff), -1.0f, 1.0f);
}

/* shader.frag:323-383,436-439; colour in internal order. MASK: 1 / 0 = the caller has tested KF_CHROMA_MASK (once for
 * the four pixels of a lane: inside the per-pixel function the wave-uniform test became a branch and two vector
 * instructions per pixel), -1 = test it here. */
template <bool RX = false, int MASK = -1, class P>
DEV f4 chroma_key(f4 c, const P& op)
{
    const f3    hsv      = rgb2hsv<RX>(f3{c.z, c.y, c.x}); /* c.bgra -> real r,g,b */
    const float hueDiff  = (0.5f - __builtin_fabsf(__builtin_fabsf(hsv.x - op.ch_target_hue) - 0.5f)) * 2.0f;
    const float satDiff  = __builtin_fminf(0.0f, op.ch_min_saturation - hsv.y);
    const float briDiff  = __builtin_fminf(0.0f, op.ch_min_brightness - hsv.z);
    const float sbScore  = __builtin_fmaxf(briDiff, satDiff);
    const float hueScore = hueDiff - op.ch_hue_width;
    const float distance = -hueScore * sbScore;
    const float dd       = distance * -2.0f + 1.0f;
    const float t        = clamp01(divc<RX>(dd - 1.0f, op.ch_soft_range, op.ch_inv_soft)); /* smoothstep(1, softness, d) */
    const float alpha    = 1.0f - t * t * (3.0f - 2.0f * t);
    if (MASK > 0 || (MASK < 0 && (op.flags & KF_CHROMA_MASK)))
        return f4{alpha, alpha, alpha, 1.0f};
    f3          sp   = hsv;
    const float diff = wrap_hue(hsv.x - op.ch_target_hue);
    const float dist = divc<RX>(__builtin_fabsf(diff), op.ch_spill, op.ch_inv_spill);
    if (dist < 1.0f) {
        sp.x = diff < 0.0f ? op.ch_target_hue - op.ch_spill : op.ch_target_hue + op.ch_spill;
        sp.y *= __builtin_fminf(1.0f, dist + op.ch_spill_saturation);
    }
    const f3 rgb = hsv2rgb(sp);
    return f4{rgb.z * alpha, rgb.y * alpha, rgb.x * alpha, alpha}; /* back to internal order */
}

/* The same for four pixels of the relaxed (16-bit) build, where a few 1e-7 do not matter (divf): value and chroma as
 * max3 / min3 (the shader's select cascade sorts the channels; only the hue still needs it), ONE wrapped hue distance
 * w for the hue score (2 |w|, the shader's 0.5 - ||h - t| - 0.5| doubled) and the spill band (|w| / spill), the
 * smoothstep argument as one multiply by -2 / softness, v_fract for fract(), value * alpha before the three channels.
 * 59 vector instructions per pixel, 79 as the compiler made them from chroma_key(). */
struct ChromaRx {
    float target, width, min_sat, min_bri, hue_lo, hue_hi, inv_spill, spill_sat, m2_inv_soft;
    template <class P>
    explicit DEV ChromaRx(const P& o)
        : target(o.ch_target_hue), width(o.ch_hue_width), min_sat(o.ch_min_saturation), min_bri(o.ch_min_brightness),
          hue_lo(o.ch_target_hue - o.ch_spill), hue_hi(o.ch_target_hue + o.ch_spill), inv_spill(o.ch_inv_spill),
          spill_sat(o.ch_spill_saturation), m2_inv_soft(-2.0f * o.ch_inv_soft)
    {
    }
};
template <bool MASK>
DEV f4 chroma_key_relaxed(f4 c, const ChromaRx& k)
{
    const float r = c.z, g = c.y, b = c.x;
    const bool  gb = !(g < b);
    const float px = gb ? g : b, py = gb ? b : g, pz = gb ? 0.0f : -1.0f, pw = gb ? -1.0f / 3.0f : 2.0f / 3.0f;
    const bool  rp = !(r < px);
    const float v = rp ? r : px, qz = rp ? pz : pw, qw = rp ? px : r;
    const float d = v - __builtin_fminf(__builtin_fminf(r, g), b);
    const float e = 1.0e-10f;
    const float h = __builtin_fabsf(__builtin_fmaf(qw - py, __builtin_amdgcn_rcpf(__builtin_fmaf(6.0f, d, e)), qz));
    const float s = d * __builtin_amdgcn_rcpf(v + e);
    const float w = wrap_hue(h - k.target), aw = __builtin_fabsf(w);
    const float hueScore = __builtin_fmaf(aw, 2.0f, -k.width);
    const float sbScore  = __builtin_fmaxf(__builtin_fminf(0.0f, k.min_bri - v), __builtin_fminf(0.0f, k.min_sat - s));
    const float t        = clamp01((hueScore * sbScore) * -k.m2_inv_soft); /* distance = -hueScore * sbScore */
    const float alpha    = __builtin_fmaf(-(t * t), __builtin_fmaf(-2.0f, t, 3.0f), 1.0f);
    if (MASK)
        return f4{alpha, alpha, alpha, 1.0f};
    const float dist = aw * k.inv_spill;
    const bool  in   = dist < 1.0f;
    const float sh   = in ? (w < 0.0f ? k.hue_lo : k.hue_hi) : h;
    const float ss   = in ? s * __builtin_fminf(1.0f, dist + k.spill_sat) : s;
    /* v * mix(1, c, s) * alpha with c = clamp(|p| - 1), p = fract(h + off) * 6 - 3, as va - (va * s) * clamp(2 - |p|):
       1 - clamp(|p| - 1) is clamp(2 - |p|) for |p| in [0, 3] */
    const float va = v * alpha, vas = va * ss;
    auto chan = [&](float hue) {
        const float p = __builtin_fmaf(__builtin_amdgcn_fractf(hue), 6.0f, -3.0f);
        return __builtin_fmaf(-vas, clamp01(2.0f - __builtin_fabsf(p)), va);
    };
    const float R = chan(sh), G = chan(sh + 2.0f / 3.0f), B = chan(sh + 1.0f / 3.0f);
    return f4{B, G, R, alpha};
}

/* shader.frag:76-86 */
template <bool RX = false, class P>
DEV float levels1(float c, const P& op)
{
    float r;
    if constexpr (RX) /* clamp((c - min) / range, 0, 1) as one fma with the clamp as its output modifier */
        r = clamp01(__builtin_fmaf(c, op.lv_inv_range, -op.lv_min_input * op.lv_inv_range));
    else
        r = __builtin_fminf(div_const(__builtin_fmaxf(c - op.lv_min_input, 0.0f), op.lv_range, op.lv_inv_range), 1.0f);
    /* pow(r, 1/gamma), r in [0,1]. gamma == 1 (levels used for their ranges only) is exactly r; the log/exp pair
       would be a few ulp off it, enough to move one 8-bit sample in ~10^5 (found by tests/test_fuzz.py). */
    if (op.lv_inv_gamma != 1.0f)
        r = __builtin_amdgcn_exp2f(op.lv_inv_gamma * __builtin_amdgcn_logf(r));
    if (op.lv_min_output == 0.0f && op.lv_max_output == 1.0f)
        return r; /* mix(0, 1, r) = 0*(1-r) + 1*r is exactly r */
    if constexpr (RX)
        return __builtin_fmaf(r, op.lv_max_output - op.lv_min_output, op.lv_min_output);
    return mixf(op.lv_min_output, op.lv_max_output, r);
}

/* levels1() on the three colours of four pixels, with its two wave-uniform questions (gamma != 1, output range other
 * than [0,1]) asked ONCE, as branches: inside the per-sample function the compiler turned both into per-sample selects
 * and evaluated the unused output mix anyway - 7 instructions per sample where the usual levels (gamma only) need 4
 * (36 cycles -> 22 by tests/ubench's rates). Same arithmetic per sample, so the 8-bit build stays bit-identical. */
template <bool RX = false, class P>
DEV void levels12(f4 col[4], const P& op)
{
    float* c[12] = {&col[0].x, &col[0].y, &col[0].z, &col[1].x, &col[1].y, &col[1].z,
                    &col[2].x, &col[2].y, &col[2].z, &col[3].x, &col[3].y, &col[3].z};
    float  off = 0.0f;
    if constexpr (RX)
        off = -op.lv_min_input * op.lv_inv_range;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        if constexpr (RX)
            *c[k] = clamp01(__builtin_fmaf(*c[k], op.lv_inv_range, off));
        else
            *c[k] = __builtin_fminf(div_const(__builtin_fmaxf(*c[k] - op.lv_min_input, 0.0f), op.lv_range, op.lv_inv_range), 1.0f);
    }
    if (op.lv_inv_gamma != 1.0f) { /* wave-uniform; the empty asm keeps it a branch */
        asm volatile("");
        float g = op.lv_inv_gamma;
        asm volatile("" : "+v"(g)); /* a VGPR: twelve uses, and an SGPR operand halves the issue rate */
#pragma unroll
        for (int k = 0; k < 12; ++k)
            *c[k] = __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(*c[k]));
    }
    if (!(op.lv_min_output == 0.0f && op.lv_max_output == 1.0f)) { /* wave-uniform and rare */
        asm volatile("");
        float lo = op.lv_min_output, hi = op.lv_max_output;
        asm volatile("" : "+v"(lo), "+v"(hi));
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            if constexpr (RX)
                *c[k] = __builtin_fmaf(*c[k], hi - lo, lo);
            else
                *c[k] = mixf(lo, hi, *c[k]);
        }
    }
}

/* shader.frag:48-70 */
template <bool RX = false, class P>
DEV void csb(f4& color, const P& op)
{
    const bool hd = op.flags & KF_IS_HD;
    const f3   L  = hd ? f3{0.0722f, 0.7152f, 0.2126f} : f3{0.114f, 0.587f, 0.299f};
    f3         c  = {color.x, color.y, color.z};
    if (color.w > 0.0f) {
        if constexpr (RX) {
            const float rw = __builtin_amdgcn_rcpf(color.w);
            c.x *= rw, c.y *= rw, c.z *= rw;
        } else {
            c.x = fdiv(c.x, color.w), c.y = fdiv(c.y, color.w), c.z = fdiv(c.z, color.w);
        }
    }
    const f3    b = {c.x * op.brt, c.y * op.brt, c.z * op.brt};
    const float i = b.x * L.x + b.y * L.y + b.z * L.z;
    const f3    s = {mixf(i, b.x, op.sat), mixf(i, b.y, op.sat), mixf(i, b.z, op.sat)};
    color.x       = mixf(0.5f, s.x, op.con) * color.w;
    color.y       = mixf(0.5f, s.y, op.con) * color.w;
    color.z       = mixf(0.5f, s.z, op.con) * color.w;
}

/* shader.frag:114-197 */
DEV f3 RGBToHSL(f3 color)
{
    f3          hsl   = {0.0f, 0.0f, 0.0f};
    const float fmin  = __builtin_fminf(__builtin_fminf(color.x, color.y), color.z);
    const float fmax  = __builtin_fmaxf(__builtin_fmaxf(color.x, color.y), color.z);
    const float delta = fmax - fmin;
    hsl.z             = (fmax + fmin) / 2.0f;
    if (delta != 0.0f) {
        hsl.y              = hsl.z < 0.5f ? delta / (fmax + fmin) : delta / (2.0f - fmax - fmin);
        const float deltaR = (((fmax - color.x) / 6.0f) + (delta / 2.0f)) / delta;
        const float deltaG = (((fmax - color.y) / 6.0f) + (delta / 2.0f)) / delta;
        const float deltaB = (((fmax - color.z) / 6.0f) + (delta / 2.0f)) / delta;
        if (color.x == fmax)
            hsl.x = deltaB - deltaG;
        else if (color.y == fmax)
            hsl.x = (1.0f / 3.0f) + deltaR - deltaB;
        else if (color.z == fmax)
            hsl.x = (2.0f / 3.0f) + deltaG - deltaR;
        if (hsl.x < 0.0f)
            hsl.x += 1.0f;
        else if (hsl.x > 1.0f)
            hsl.x -= 1.0f;
    }
    return hsl;
}
DEV float HueToRGB(float f1, float f2, float hue)
{
    if (hue < 0.0f)
        hue += 1.0f;
    else if (hue > 1.0f)
        hue -= 1.0f;
    if ((6.0f * hue) < 1.0f)
        return f1 + (f2 - f1) * 6.0f * hue;
    if ((2.0f * hue) < 1.0f)
        return f2;
    if ((3.0f * hue) < 2.0f)
        return f1 + (f2 - f1) * ((2.0f / 3.0f) - hue) * 6.0f;
    return f1;
}
DEV f3 HSLToRGB(f3 hsl)
{
    if (hsl.y == 0.0f)
        return f3{hsl.z, hsl.z, hsl.z};
    const float f2 = hsl.z < 0.5f ? hsl.z * (1.0f + hsl.y) : (hsl.z + hsl.y) - (hsl.y * hsl.z);
    const float f1 = 2.0f * hsl.z - f2;
    return f3{HueToRGB(f1, f2, hsl.x + (1.0f / 3.0f)), HueToRGB(f1, f2, hsl.x), HueToRGB(f1, f2, hsl.x - (1.0f / 3.0f))};
}

/* shader.frag:208-223 */
DEV float BlendAddf(float b, float s) { return __builtin_fminf(b + s, 1.0f); }
DEV float BlendSubstractf(float b, float s) { return __builtin_fmaxf(b + s - 1.0f, 0.0f); }
DEV float BlendLinearLightf(float b, float s)
{
    return s < 0.5f ? BlendSubstractf(b, 2.0f * s) : BlendAddf(b, 2.0f * (s - 0.5f));
}
DEV float BlendScreenf(float b, float s) { return 1.0f - ((1.0f - b) * (1.0f - s)); }
DEV float BlendOverlayf(float b, float s)
{
    return b < 0.5f ? (2.0f * b * s) : (1.0f - 2.0f * (1.0f - b) * (1.0f - s));
}
DEV float BlendColorDodgef(float b, float s) { return s == 1.0f ? s : __builtin_fminf(b / (1.0f - s), 1.0f); }
DEV float BlendColorBurnf(float b, float s) { return s == 0.0f ? s : __builtin_fmaxf(1.0f - ((1.0f - b) / s), 0.0f); }
DEV float BlendVividLightf(float b, float s)
{
    return s < 0.5f ? BlendColorBurnf(b, 2.0f * s) : BlendColorDodgef(b, 2.0f * (s - 0.5f));
}
DEV float BlendPinLightf(float b, float s)
{
    return s < 0.5f ? __builtin_fminf(2.0f * s, b) : __builtin_fmaxf(2.0f * (s - 0.5f), b);
}
DEV float BlendHardMixf(float b, float s) { return BlendVividLightf(b, s) < 0.5f ? 0.0f : 1.0f; }
DEV float BlendReflectf(float b, float s) { return s == 1.0f ? s : __builtin_fminf(b * b / (1.0f - s), 1.0f); }

#define B3(f) f3{f(b.x, s.x), f(b.y, s.y), f(b.z, s.z)}
#define B3R(f) f3{f(s.x, b.x), f(s.y, b.y), f(s.z, b.z)}

/* shader.frag:387-422 with its aliasing kept: 12 (soft_light) and >= 29 (mix) fall through to normal,
 * 25 ("contrast") runs BlendHue. b = back, s = fore. The mode is a template parameter; get_blend_color() below
 * picks the body with a SCALAR switch. When the switch itself sat in the out-of-line function, the mode arrived in a
 * VGPR (function ABI) and the switch was lowered as divergent control flow: ~170 exec-mask blocks walked per pixel.
 * (The bodies cannot be inlined into the op loops instead: with 28 more paths to the loop latch the compiler no
 * longer proves the op fields uniform and refuses the scalar loads.) */
template <int mode>
DEV f3 blend_color(f3 b, f3 s)
{
    switch (mode) {
        case 1: return f3{__builtin_fmaxf(s.x, b.x), __builtin_fmaxf(s.y, b.y), __builtin_fmaxf(s.z, b.z)};
        case 2: return f3{__builtin_fminf(s.x, b.x), __builtin_fminf(s.y, b.y), __builtin_fminf(s.z, b.z)};
        case 3: return f3{b.x * s.x, b.y * s.y, b.z * s.z};
        case 4: return f3{(b.x + s.x) / 2.0f, (b.y + s.y) / 2.0f, (b.z + s.z) / 2.0f};
        case 5:
        case 16: return B3(BlendAddf);
        case 6:
        case 17: return B3(BlendSubstractf);
        case 7: return f3{__builtin_fabsf(b.x - s.x), __builtin_fabsf(b.y - s.y), __builtin_fabsf(b.z - s.z)};
        case 8:
            return f3{1.0f - __builtin_fabsf(1.0f - b.x - s.x), 1.0f - __builtin_fabsf(1.0f - b.y - s.y),
                      1.0f - __builtin_fabsf(1.0f - b.z - s.z)};
        case 9:
            return f3{b.x + s.x - 2.0f * b.x * s.x, b.y + s.y - 2.0f * b.y * s.y, b.z + s.z - 2.0f * b.z * s.z};
        case 10: return B3(BlendScreenf);
        case 11: return B3(BlendOverlayf);
        case 13: return B3R(BlendOverlayf);
        case 14: return B3(BlendColorDodgef);
        case 15: return B3(BlendColorBurnf);
        case 18: return B3(BlendLinearLightf);
        case 19: return B3(BlendVividLightf);
        case 20: return B3(BlendPinLightf);
        case 21: return B3(BlendHardMixf);
        case 22: return B3(BlendReflectf);
        case 23: return B3R(BlendReflectf);
        case 24:
            return f3{__builtin_fminf(b.x, s.x) - __builtin_fmaxf(b.x, s.x) + 1.0f,
                      __builtin_fminf(b.y, s.y) - __builtin_fmaxf(b.y, s.y) + 1.0f,
                      __builtin_fminf(b.z, s.z) - __builtin_fmaxf(b.z, s.z) + 1.0f};
        case 25: {
            const f3 h = RGBToHSL(b);
            return HSLToRGB(f3{RGBToHSL(s).x, h.y, h.z});
        }
        case 26: {
            const f3 h = RGBToHSL(b);
            return HSLToRGB(f3{h.x, RGBToHSL(s).y, h.z});
        }
        case 27: {
            const f3 h = RGBToHSL(s);
            return HSLToRGB(f3{h.x, h.y, RGBToHSL(b).z});
        }
        case 28: {
            const f3 h = RGBToHSL(b);
            return HSLToRGB(f3{h.x, h.y, RGBToHSL(s).z});
        }
        default: return s;
    }
}

/* f(std::integral_constant<int, m>) for the executed mode m (0 for the modes that alias to normal); mode is
 * wave-uniform */
template <class F>
DEV void dispatch_mode(int mode, F&& f)
{
    switch (__builtin_amdgcn_readfirstlane(mode)) {
#define M(k) case k: f(std::integral_constant<int, k>{}); break;
        M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(21)
        M(22) M(23) M(24) M(25) M(26) M(27) M(28)
#undef M
        default: f(std::integral_constant<int, 0>{}); break;
    }
}

/* run-time mode, out of line */
__device__ __noinline__ f3 get_blend_color(int mode, f3 b, f3 s)
{
    f3 r = s;
    dispatch_mode(mode, [&](auto m) { r = blend_color<decltype(m)::value>(b, s); });
    return r;
}

/* the same for a lane's four pixels with ONE call: after a call the caller reloads the SGPRs it held from memory,
 * and at the three waves per SIMD of the kernel variant that blends, four such stalls per op are not hidden.
 * The 12 + 12 floats travel as vector types: those are passed and returned in VGPRs, while a struct of this size
 * goes through scratch memory (byval / sret), which cost more than the three calls saved (16 blend layers at
 * 2160p: 0.79 ms with four calls, 0.87 ms with the struct). */
typedef float v12f __attribute__((ext_vector_type(12)));
__device__ __noinline__ v12f get_blend_color4(int mode, v12f b, v12f s)
{
    v12f r = s;
    dispatch_mode(mode, [&](auto m) {
        constexpr int M = decltype(m)::value;
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const f3      c = blend_color<M>(f3{b[3 * i], b[3 * i + 1], b[3 * i + 2]}, f3{s[3 * i], s[3 * i + 1], s[3 * i + 2]});
            r[3 * i] = c.x, r[3 * i + 1] = c.y, r[3 * i + 2] = c.z;
        });
    });
    return r;
}

/* shader.frag:474-521 from GL texels to the internal (b,g,r,a) vector */
DEV f4 swizzle_format(int pf, f4 a)
{
    switch (pf) {
        case 0: return f4{a.x, a.x, a.x, 1.0f};
        case 1: return f4{a.z, a.y, a.x, a.w};
        case 2: return a;
        case 3: return f4{a.w, a.x, a.y, a.z};
        case 4: return f4{a.y, a.z, a.w, a.x};
        case 7: {
            const float l = divc<RXF>(a.x - 0.065f, 0.859f, 1.0f / 0.859f);
            return f4{l, l, l, 1.0f};
        }
        case 8: return f4{a.z, a.y, a.x, 1.0f};
        case 9: return f4{a.x, a.y, a.z, 1.0f};
        default: return f4{0.0f, 0.0f, 0.0f, 0.0f};
    }
}

/* source colour of a KOP_DRAW_PLANES at normalised (s,t) */
DEV f4 source_color_norm(const KOp& op, float s, float t)
{
    const bool hd = op.flags & KF_IS_HD;
    switch (op.pixel_format) {
        case 5:
            return ycbcra_to_rgba(sample_norm(op.planes[0], s, t).x, sample_norm(op.planes[1], s, t).x,
                                  sample_norm(op.planes[2], s, t).x, 1.0f, hd);
        case 6:
            return ycbcra_to_rgba(sample_norm(op.planes[0], s, t).x, sample_norm(op.planes[1], s, t).x,
                                  sample_norm(op.planes[2], s, t).x, sample_norm(op.planes[3], s, t).x, hd);
        case 10: {
            const float y = sample_norm(op.planes[0], s, t).y;
            const f4    c = sample_norm(op.planes[1], s, t);
            return ycbcra_to_rgba(y, c.z, c.x, 1.0f, hd);
        }
        default: return swizzle_format(op.pixel_format, sample_norm(op.planes[0], s, t));
    }
}

/* same, KF_UNIT_SCALE addressing: (px,py) = plane-0 texel */
DEV f4 source_color_unit(const KOp& op, int px, int py)
{
    const bool   hd = op.flags & KF_IS_HD;
    const float  w0 = op.planes[0].fw, h0 = op.planes[0].fh;
#define RX(k) (op.planes[k].fw / w0)
#define RY(k) (op.planes[k].fh / h0)
    switch (op.pixel_format) {
        case 5:
            return ycbcra_to_rgba(sample_unit(op.planes[0], px, py, 1.0f, 1.0f).x,
                                  sample_unit(op.planes[1], px, py, RX(1), RY(1)).x,
                                  sample_unit(op.planes[2], px, py, RX(2), RY(2)).x, 1.0f, hd);
        case 6:
            return ycbcra_to_rgba(sample_unit(op.planes[0], px, py, 1.0f, 1.0f).x,
                                  sample_unit(op.planes[1], px, py, RX(1), RY(1)).x,
                                  sample_unit(op.planes[2], px, py, RX(2), RY(2)).x,
                                  sample_unit(op.planes[3], px, py, RX(3), RY(3)).x, hd);
        case 10: {
            const float y = sample_unit(op.planes[0], px, py, 1.0f, 1.0f).y;
            const f4    c = sample_unit(op.planes[1], px, py, RX(1), RY(1));
            return ycbcra_to_rgba(y, c.z, c.x, 1.0f, hd);
        }
        default: return swizzle_format(op.pixel_format, sample_unit(op.planes[0], px, py, 1.0f, 1.0f));
    }
#undef RX
#undef RY
}

/* shader.frag:523-541 after get_rgba_color: adjustments, keys, opacity, invert, blend; returns the
 * UNQUANTISED result in internal order. back = current content of the destination surface. */
/* shader.frag:526-531: chroma key, levels, CSB on the source colour */
DEV f4 shade_adjust(const KOp& op, f4 color)
{
    if (op.flags & KF_CHROMA) {
        if constexpr (RXF) { /* the arithmetic of the 4-pixel kernels: the same layer gives the same pixels in either */
            const ChromaRx k(op);
            color = (op.flags & KF_CHROMA_MASK) ? chroma_key_relaxed<true>(color, k) : chroma_key_relaxed<false>(color, k);
        } else {
            color = chroma_key<false>(color, op);
        }
    }
    if (op.flags & KF_LEVELS) {
        color.x = levels1<RXF>(color.x, op);
        color.y = levels1<RXF>(color.y, op);
        color.z = levels1<RXF>(color.z, op);
    }
    if (op.flags & KF_CSB)
        csb<RXF>(color, op);
    return color;
}

/* shader.frag:532-541: keys, opacity, invert, blend. No FMA contraction in here: llvmpipe evaluates these
 * unfused, and with "nice" opacities (0.5, 0.25 ...) the UNORM store that follows hits exact .5 ties that only
 * the same sequence of roundings resolves the same way. Returns the UNQUANTISED result, internal order. */
enum : int { MODE_RUNTIME = -2, MODE_NONE = -1 }; /* or the blend mode itself, inline */
struct BlendArgs { /* what shade_blend reads of an op, by value */
    uint32_t flags;
    int32_t  local_key, layer_key;
    float    opacity;
    int32_t  blend_mode;
};
template <int MODE, class P>
DEV f4 shade_blend(const P& op, f4 color, float lkey, float ykey, f4 back)
{
    /* The optional steps are made unconditional with neutral operands instead of wave-uniform branches (which
       the compiler turns into compute-both-and-select): x*1 == x, fma(x, 1, 0) == x and fma(x, -1, 1) == 1 - x
       exactly, so the roundings are those of the shader's own sequence. */
    const float k1 = op.local_key != KS_NONE ? lkey : 1.0f;
    const float k2 = op.layer_key != KS_NONE ? ykey : 1.0f;
    color.x *= k1, color.y *= k1, color.z *= k1, color.w *= k1;
    color.x *= k2, color.y *= k2, color.z *= k2, color.w *= k2;
    color.x *= op.opacity, color.y *= op.opacity, color.z *= op.opacity, color.w *= op.opacity;
    const bool  inv = op.flags & KF_INVERT;
    const float im = inv ? -1.0f : 1.0f, ia0 = inv ? 1.0f : 0.0f;
    f4 fore = {__builtin_fmaf(color.x, im, ia0), __builtin_fmaf(color.y, im, ia0), __builtin_fmaf(color.z, im, ia0),
               __builtin_fmaf(color.w, im, ia0)};
    bool blend = MODE >= 0;
    if constexpr (MODE == MODE_RUNTIME)
        blend = op.blend_mode != 0;
    if (blend) { /* shader.frag:427-428 */
        /* The two un-premultiply divides are correctly rounded in BOTH builds: they are where the shader amplifies
           (rgb / alpha with alpha down to one code), and only blend-mode ops pay for them. With the 1-ulp a * rcp(b) of
           the relaxed build it was unproven whether a 16-bit residual of 3 - 10 LSB16 at small alphas was the scene's
           or this arithmetic's. */
        const float ba = back.w + 0.0000001f, fa = fore.w + 0.0000001f;
        const f3    B = {fdiv(back.x, ba), fdiv(back.y, ba), fdiv(back.z, ba)};
        const f3    F = {fdiv(fore.x, fa), fdiv(fore.y, fa), fdiv(fore.z, fa)};
        f3          bc;
        if constexpr (MODE == MODE_RUNTIME)
            bc = get_blend_color(op.blend_mode, B, F);
        else
            bc = blend_color<(MODE >= 0 ? MODE : 0)>(B, F);
        fore.x = bc.x * fore.w, fore.y = bc.y * fore.w, fore.z = bc.z * fore.w;
    }
    /* shader.frag:429-433: additive keyer = fore + back, linear = fore + (1 - fore.a) * back */
    const float ia = (op.flags & KF_ADDITIVE) ? 1.0f : 1.0f - fore.w;
    return f4{fore.x + ia * back.x, fore.y + ia * back.y, fore.z + ia * back.z, fore.w + ia * back.w};
}

/* shade_blend<MODE_RUNTIME> with a non-zero mode for four pixels, operation for operation, around one call of
 * get_blend_color4; results quantised into D */
template <int BITS>
DEV void shade_blend_mode4(const BlendArgs op, const f4 col[4], f4 D[4], const float K0[4], const float K1[4])
{
    const bool  inv = op.flags & KF_INVERT;
    const float im = inv ? -1.0f : 1.0f, ia0 = inv ? 1.0f : 0.0f;
    f4          fore[4];
    f3          B[4], F[4];
    static_for<0, 4>([&](auto I) {
        constexpr int i  = decltype(I)::value;
        const float   k1 = op.local_key != KS_NONE ? (op.local_key == KS_KEY1 ? K1[i] : K0[i]) : 1.0f;
        const float   k2 = op.layer_key != KS_NONE ? (op.layer_key == KS_KEY1 ? K1[i] : K0[i]) : 1.0f;
        f4            c  = col[i];
        c.x *= k1, c.y *= k1, c.z *= k1, c.w *= k1;
        c.x *= k2, c.y *= k2, c.z *= k2, c.w *= k2;
        c.x *= op.opacity, c.y *= op.opacity, c.z *= op.opacity, c.w *= op.opacity;
        fore[i] = f4{__builtin_fmaf(c.x, im, ia0), __builtin_fmaf(c.y, im, ia0), __builtin_fmaf(c.z, im, ia0),
                     __builtin_fmaf(c.w, im, ia0)};
        const float ba = D[i].w + 0.0000001f, fa = fore[i].w + 0.0000001f; /* exact divides in both builds: see shade_blend */
        B[i] = f3{fdiv(D[i].x, ba), fdiv(D[i].y, ba), fdiv(D[i].z, ba)};
        F[i] = f3{fdiv(fore[i].x, fa), fdiv(fore[i].y, fa), fdiv(fore[i].z, fa)};
    });
    const v12f r = get_blend_color4(op.blend_mode,
                                    v12f{B[0].x, B[0].y, B[0].z, B[1].x, B[1].y, B[1].z, B[2].x, B[2].y, B[2].z, B[3].x, B[3].y, B[3].z},
                                    v12f{F[0].x, F[0].y, F[0].z, F[1].x, F[1].y, F[1].z, F[2].x, F[2].y, F[2].z, F[3].x, F[3].y, F[3].z});
    const f3 bc[4] = {f3{r[0], r[1], r[2]}, f3{r[3], r[4], r[5]}, f3{r[6], r[7], r[8]}, f3{r[9], r[10], r[11]}};
    static_for<0, 4>([&](auto I) {
        constexpr int i  = decltype(I)::value;
        const float   fw = fore[i].w;
        const f3      f  = {bc[i].x * fw, bc[i].y * fw, bc[i].z * fw};
        const float   ia = (op.flags & KF_ADDITIVE) ? 1.0f : 1.0f - fw;
        D[i] = f4{quantize<BITS>(f.x + ia * D[i].x), quantize<BITS>(f.y + ia * D[i].y), quantize<BITS>(f.z + ia * D[i].z),
                  quantize<BITS>(fw + ia * D[i].w)};
    });
}

/* the whole of shader.frag:523-541 */
DEV f4 shade(const KOp& op, f4 color, float lkey, float ykey, f4 back)
{
    return shade_blend<MODE_RUNTIME>(op, shade_adjust(op, color), lkey, ykey, back);
}

/* ------------------------------------------------------------------ the compositor */

DEV f4 sel4(bool c, f4 a, f4 b) { return f4{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w}; }

/* ---- straight-line bilinear bodies for the two common 8-bit layouts (KSampleClass) ---- */

/* == one byte of lerp8x4. |b - a| and w fit in 9 bits, so the 24-bit multiply-add is exact; written out because the
 * compiler only proves that for operands straight from a byte load and made the second-level lerp a 64-bit
 * multiply-add (v_mad_u64_u32) */
DEV int lerp8s(int a, int b, int w)
{
    int t;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t) : "v"(w), "v"(b - a), "s"(128));
    return a + (t >> 8);
}

/* byte offsets of the four taps of a GL_LINEAR + CLAMP_TO_EDGE fetch at 24.8 fixed-point texel coordinates
 * (already shifted by half a texel), and the two 8-bit weights */
struct Taps {
    uint32_t o0, o1; /* byte offsets of the tap PAIR (columns x, x + 1) in row y0 and in row y1 */
    int      wx, wy;
};
/* The two column taps of a row are neighbours in memory, so one load of two texels serves both (planes of this class
 * are at least two texels wide, cghip_plan.cpp: classify_sampling). CLAMP_TO_EDGE: left of the plane both taps are
 * texel 0 - weight 0 on the pair (0, 1); right of column w - 2 both are texel w - 1 - weight 256 on the pair
 * (w - 2, w - 1): lerp8 with weight 256 returns its second operand exactly. Twelve loads per pixel of a planar source
 * became six. */
DEV Taps make_taps(int iu, int iv, int w, int h, uint32_t texel_bytes)
{
    const int x0 = iu >> 8;
    int       y0 = iv >> 8, wx = iu & 0xff;
    wx           = x0 < 0 ? 0 : wx;
    wx           = x0 > w - 2 ? 256 : wx;
    const int xp = iclamp(x0, 0, w - 2), y1 = iclamp(y0 + 1, 0, h - 1);
    y0           = iclamp(y0, 0, h - 1);
    const uint32_t r0 = (uint32_t)y0 * (uint32_t)w, r1 = (uint32_t)y1 * (uint32_t)w;
    return Taps{(r0 + (uint32_t)xp) * texel_bytes, (r1 + (uint32_t)xp) * texel_bytes, wx, iv & 0xff};
}
/* 1-channel plane: the filtered code 0..255 (x first, then y, 8-bit intermediates: sample8_fixed for one byte;
 * a zero weight leaves the first operand untouched, so no tap is conditional) */
DEV float tap1(uint64_t ptr, const Taps t)
{
    const GB  b   = (GB)ptr;
    const int top = load_at<uint16_t>(b, t.o0), bot = load_at<uint16_t>(b, t.o1);
    return (float)lerp8s(lerp8s(top & 0xff, top >> 8, t.wx), lerp8s(bot & 0xff, bot >> 8, t.wx), t.wy);
}
/* 4-channel plane: the four filtered bytes, memory order */
DEV uint32_t tap4(uint64_t ptr, const Taps t)
{
    const GB    b   = (GB)ptr;
    const u32x2 top = load_at<u32x2>(b, t.o0), bot = load_at<u32x2>(b, t.o1);
    return lerp8x4(lerp8x4(top.x, top.y, (uint32_t)t.wx), lerp8x4(bot.x, bot.y, (uint32_t)t.wx), (uint32_t)t.wy);
}

/* op fields of the general kernel, fetched in one scalar batch at the top of the op loop (see HotOp) */
struct HotG {
    int32_t  kind;
    uint32_t flags;
    int32_t  dst, src_surface, local_key, layer_key, pixel_format;
    int32_t  x0, y0, x1, y1, unit_dx, unit_dy, sample_class;
    uint64_t ptr0, ptr1, ptr2, ptr3;
    int32_t  w0, h0, w1, h1;
    float    fw0, fh0, fw1, fh1;
    int32_t  bits0;
    int32_t  fused, ax0, ay0, ax1, ay1; /* KOp::fused: x0..y1 is the composite's box, ax0..ay1 the item's */
};
template <bool WITH_FUSED = true> /* false: the sampled bodies of the fast kernel, which take the item's columns from HotOp */
DEV HotG load_hot_g(const KOp& op)
{
    HotG h = {op.kind, op.flags, op.dst, op.src_surface, op.local_key, op.layer_key, op.pixel_format, op.x0, op.y0, op.x1,
              op.y1, op.unit_dx, op.unit_dy, op.sample_class, (uint64_t)op.planes[0].ptr, (uint64_t)op.planes[1].ptr,
              (uint64_t)op.planes[2].ptr, (uint64_t)op.planes[3].ptr, op.planes[0].w, op.planes[0].h, op.planes[1].w,
              op.planes[1].h, op.planes[0].fw, op.planes[0].fh, op.planes[1].fw, op.planes[1].fh, op.planes[0].bits,
              0, 0, 0, 0, 0};
    asm volatile(""
                 : "+s"(h.kind), "+s"(h.flags), "+s"(h.dst), "+s"(h.src_surface), "+s"(h.local_key), "+s"(h.layer_key),
                   "+s"(h.pixel_format), "+s"(h.x0), "+s"(h.y0), "+s"(h.x1), "+s"(h.y1), "+s"(h.unit_dx), "+s"(h.unit_dy),
                   "+s"(h.sample_class), "+s"(h.ptr0), "+s"(h.ptr1), "+s"(h.ptr2), "+s"(h.ptr3), "+s"(h.w0), "+s"(h.h0),
                   "+s"(h.w1), "+s"(h.h1), "+s"(h.fw0), "+s"(h.fh0), "+s"(h.fw1), "+s"(h.fh1), "+s"(h.bits0));
    if constexpr (WITH_FUSED) { /* the item's own box is fetched where a fused op needs it: unfused ops do not pay */
        h.fused = op.fused;
        asm volatile("" : "+s"(h.fused));
    }
    return h;
}

/* SC_PLANAR8 / SC_PACKED8 source colour. Texel coordinates of a plane of size (fw, fh): normalised (s,t) give
 * rint(s*fw*256) - 128 (sample_norm); at unit scale texel (px,py) of plane 0 gives rint((px+.5)*(fw/fw0)*256) - 128
 * (sample_unit). */
template <bool UNIT>
DEV f4 source_color_class(const KOp& op, const HotG& h, float s, float t, int px, int py)
{
    auto coord = [&](float v, float size, float size0, int p) -> int {
        if (UNIT)
            return (int)__builtin_rintf(((float)p + 0.5f) * ((size / size0) * 256.0f)) - 128;
        return (int)__builtin_rintf(v * size * 256.0f) - 128;
    };
    const Taps t0 = make_taps(coord(s, h.fw0, h.fw0, px), coord(t, h.fh0, h.fh0, py), h.w0, h.h0,
                              h.sample_class == SC_PACKED8 ? 4u : h.sample_class == SC_UYVY8 ? 2u : 1u);
    if (h.sample_class == SC_UYVY8) { /* shader.frag:513-519: Y = plane0.g, Cb = plane1.b, Cr = plane1.r */
        const Taps     t1 = make_taps(coord(s, h.fw1, h.fw0, px), coord(t, h.fh1, h.fh0, py), h.w1, h.h1, 4u);
        const uint32_t yt = load_at<uint32_t>((GB)h.ptr0, t0.o0), yb = load_at<uint32_t>((GB)h.ptr0, t0.o1);
        const u32x2    ct = load_at<u32x2>((GB)h.ptr1, t1.o0), cb = load_at<u32x2>((GB)h.ptr1, t1.o1);
        const float    sc = 1.0f / 255.0f;
        const float    Y  = (float)lerp8s(lerp8s((yt >> 8) & 0xff, yt >> 24, t0.wx), lerp8s((yb >> 8) & 0xff, yb >> 24, t0.wx), t0.wy);
        const float    U  = (float)lerp8s(lerp8s(ct.x & 0xff, ct.y & 0xff, t1.wx), lerp8s(cb.x & 0xff, cb.y & 0xff, t1.wx), t1.wy);
        const float    V  = (float)lerp8s(lerp8s((ct.x >> 16) & 0xff, (ct.y >> 16) & 0xff, t1.wx),
                                          lerp8s((cb.x >> 16) & 0xff, (cb.y >> 16) & 0xff, t1.wx), t1.wy);
        return ycbcra_to_rgba(Y * sc, U * sc, V * sc, 1.0f, h.flags & KF_IS_HD);
    }
    if (h.sample_class == SC_PACKED8) {
        const int      pf  = h.pixel_format;
        const uint32_t sel = pf == 1 ? 0x03020100u : pf == 2 ? 0x03000102u : pf == 3 ? 0x00010203u : 0x02030001u;
        const uint32_t m   = tap4(h.ptr0, t0);
        const uint32_t u   = __builtin_amdgcn_perm(m, m, sel);
        const float    sc  = 1.0f / 255.0f;
        return f4{(float)(u & 0xff) * sc, (float)((u >> 8) & 0xff) * sc, (float)((u >> 16) & 0xff) * sc,
                  (float)(u >> 24) * sc};
    }
    const float sc = 1.0f / 255.0f;
    const Taps  t1 = make_taps(coord(s, h.fw1, h.fw0, px), coord(t, h.fh1, h.fh0, py), h.w1, h.h1, 1u);
    const float Y = tap1(h.ptr0, t0) * sc, Cb = tap1(h.ptr1, t1) * sc, Cr = tap1(h.ptr2, t1) * sc;
    float       A = 1.0f;
    if (h.pixel_format == 6)
        A = tap1(h.ptr3, t0) * sc;
    return ycbcra_to_rgba(Y, Cb, Cr, A, h.flags & KF_IS_HD);
}

/* ---- the same for the two common 16-bit layouts: fp32 weights (sample16_float), straight-line ---- */
struct TapsF {
    uint32_t o00, o01, o10, o11;
    float    wx, wy;
};
DEV TapsF make_taps_f(float u, float v, int w, int h, uint32_t texel_bytes)
{
    const float fx = __builtin_floorf(u), fy = __builtin_floorf(v);
    int         x0 = (int)fx, y0 = (int)fy;
    const int   x1 = iclamp(x0 + 1, 0, w - 1), y1 = iclamp(y0 + 1, 0, h - 1);
    x0             = iclamp(x0, 0, w - 1);
    y0             = iclamp(y0, 0, h - 1);
    const uint32_t r0 = (uint32_t)y0 * (uint32_t)w, r1 = (uint32_t)y1 * (uint32_t)w;
    return TapsF{(r0 + (uint32_t)x0) * texel_bytes, (r0 + (uint32_t)x1) * texel_bytes, (r1 + (uint32_t)x0) * texel_bytes,
                 (r1 + (uint32_t)x1) * texel_bytes, u - fx, v - fy};
}
/* v0 + w*(v1 - v0), x first, then y; a zero weight leaves the first operand untouched, so no tap is conditional */
DEV float bilerp_f(float a00, float a01, float a10, float a11, float wx, float wy)
{
    const float a = a00 + wx * (a01 - a00), c = a10 + wx * (a11 - a10);
    return a + wy * (c - a);
}
/* 1-channel 16-bit plane: tap pairs again (see make_taps); at the plane's edges, where CLAMP_TO_EDGE makes both taps
 * the same texel, a byte permute overwrites the pair's other half with it (a + w * (b - a) with b == a is a) */
struct TapsP {
    uint32_t o0, o1, sel;
    float    wx, wy;
};
DEV TapsP make_taps_p(float u, float v, int w, int h)
{
    const float fx = __builtin_floorf(u), fy = __builtin_floorf(v);
    const int   x0 = (int)fx;
    int         y0 = (int)fy;
    const int   y1 = iclamp(y0 + 1, 0, h - 1);
    y0             = iclamp(y0, 0, h - 1);
    const uint32_t xp = (uint32_t)iclamp(x0, 0, w - 2);
    const uint32_t sel = x0 < 0 ? 0x01000100u : (x0 > w - 2 ? 0x03020302u : 0x03020100u);
    return TapsP{((uint32_t)y0 * (uint32_t)w + xp) * 2u, ((uint32_t)y1 * (uint32_t)w + xp) * 2u, sel, u - fx, v - fy};
}
DEV float tap1_16(uint64_t ptr, const TapsP t)
{
    const GB       b  = (GB)ptr;
    const float    sc = 1.0f / 65535.0f;
    const uint32_t top = load_at<uint32_t>(b, t.o0), bot = load_at<uint32_t>(b, t.o1);
    const uint32_t tp = __builtin_amdgcn_perm(top, top, t.sel), bp = __builtin_amdgcn_perm(bot, bot, t.sel);
    return bilerp_f((float)(tp & 0xffff) * sc, (float)(tp >> 16) * sc, (float)(bp & 0xffff) * sc, (float)(bp >> 16) * sc, t.wx, t.wy);
}
template <bool UNIT>
DEV f4 source_color_class16(const KOp& op, const HotG& h, float s, float t, int px, int py)
{
    auto coord = [&](float v, float size, float size0, int p) -> float {
        if (UNIT)
            return ((float)p + 0.5f) * (size / size0) - 0.5f; /* sample_unit */
        return v * size - 0.5f;                               /* sample_norm */
    };
    const float sc = 1.0f / 65535.0f;
    if (h.sample_class == SC_PACKED16) {
        const TapsF t0 = make_taps_f(coord(s, h.fw0, h.fw0, px), coord(t, h.fh0, h.fh0, py), h.w0, h.h0, 8u);
        const GB    b  = (GB)h.ptr0;
        const u32x2 a00 = load_at<u32x2>(b, t0.o00), a01 = load_at<u32x2>(b, t0.o01);
        const u32x2 a10 = load_at<u32x2>(b, t0.o10), a11 = load_at<u32x2>(b, t0.o11);
#define CH(v, k) ((float)((k) == 0 ? (v).x & 0xffff : (k) == 1 ? (v).x >> 16 : (k) == 2 ? (v).y & 0xffff : (v).y >> 16) * sc)
#define LERP(k) bilerp_f(CH(a00, k), CH(a01, k), CH(a10, k), CH(a11, k), t0.wx, t0.wy)
        const float m0 = LERP(0), m1 = LERP(1), m2 = LERP(2), m3 = LERP(3);
#undef LERP
#undef CH
        return swizzle_format(h.pixel_format, f4{m2, m1, m0, m3}); /* gl_texel of a GL_BGRA upload */
    }
    const TapsP t0 = make_taps_p(coord(s, h.fw0, h.fw0, px), coord(t, h.fh0, h.fh0, py), h.w0, h.h0);
    const TapsP t1 = make_taps_p(coord(s, h.fw1, h.fw0, px), coord(t, h.fh1, h.fh0, py), h.w1, h.h1);
    const float Y = tap1_16(h.ptr0, t0), Cb = tap1_16(h.ptr1, t1), Cr = tap1_16(h.ptr2, t1);
    float       A = 1.0f;
    if (h.pixel_format == 6)
        A = tap1_16(h.ptr3, t0);
    return ycbcra_to_rgba(Y, Cb, Cr, A, h.flags & KF_IS_HD);
}

/* Source colour of a KOP_DRAW_PLANES at pixel (x,y); false when the pixel centre is outside the quad. */
DEV bool source_color(const KOp& op, const HotG& h, int x, int y, f4& color)
{
    /* 1:1 sources are addressed by texel index in BOTH kernels, whatever the bit depth. For 16-bit sources the
       reference pipeline under llvmpipe filters with fp32 weights from coordinates a few ulp off the texel centre,
       i.e. mixes in ~1e-6 of a neighbour (1 LSB16 on ~1 % of samples; GPUs with fixed-point weights do not): taking
       the exact texel keeps this kernel and cghip_composite_fast identical on the layers they share. */
    if (h.flags & KF_UNIT_SCALE) {
        const int px = x + h.unit_dx, py = y + h.unit_dy;
        color = ((h.sample_class == SC_PLANAR16) | (h.sample_class == SC_PACKED16)) ? source_color_class16<true>(op, h, 0.0f, 0.0f, px, py)
                : h.sample_class != SC_GENERIC ? source_color_class<true>(op, h, 0.0f, 0.0f, px, py)
                                               : source_color_unit(op, px, py);
        return true;
    }
    const double dx = (double)x, dy = (double)y;
    bool         in0 = true;
    float        s, t;
    if (!(h.flags & KF_RECT)) { /* which triangle covers the pixel centre, if any */
        /* all six edge functions, then the tests: with short-circuit && every edge sat in a block of its own behind
           its own scalar load and wait (six serial scalar-cache round trips per op and wave) */
        double e0[3], e1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            e0[k] = __builtin_fma(op.tri[0].e[k][0], dx, __builtin_fma(op.tri[0].e[k][1], dy, op.tri[0].e[k][2]));
            e1[k] = __builtin_fma(op.tri[1].e[k][0], dx, __builtin_fma(op.tri[1].e[k][1], dy, op.tri[1].e[k][2]));
        }
        in0            = (e0[0] >= 0.0) & (e0[1] >= 0.0) & (e0[2] >= 0.0);
        const bool in1 = (op.ntri > 1) & (e1[0] >= 0.0) & (e1[1] >= 0.0) & (e1[2] >= 0.0);
        if (!in0 && !in1)
            return false;
    }
    /* the varyings of both triangles are evaluated and the results selected: selecting the 9 fp64 coefficients per
       pixel costs more than the second evaluation */
    {
        const KTri& T0 = op.tri[0];
        float s0 = (float)__builtin_fma(T0.a[0][1], dx, __builtin_fma(T0.a[0][2], dy, T0.a[0][0]));
        float t0 = (float)__builtin_fma(T0.a[1][1], dx, __builtin_fma(T0.a[1][2], dy, T0.a[1][0]));
        float q0 = 1.0f;
        if (h.flags & KF_PROJECTIVE)
            q0 = (float)__builtin_fma(T0.a[2][1], dx, __builtin_fma(T0.a[2][2], dy, T0.a[2][0]));
        if (!(h.flags & KF_RECT)) {
            const KTri& T1 = op.tri[1];
            const float s1 = (float)__builtin_fma(T1.a[0][1], dx, __builtin_fma(T1.a[0][2], dy, T1.a[0][0]));
            const float t1 = (float)__builtin_fma(T1.a[1][1], dx, __builtin_fma(T1.a[1][2], dy, T1.a[1][0]));
            s0 = in0 ? s0 : s1, t0 = in0 ? t0 : t1;
            if (h.flags & KF_PROJECTIVE) {
                const float q1 = (float)__builtin_fma(T1.a[2][1], dx, __builtin_fma(T1.a[2][2], dy, T1.a[2][0]));
                q0             = in0 ? q0 : q1;
            }
        }
        s = s0, t = t0;
        if (h.flags & KF_PROJECTIVE) {
            s /= q0;
            t /= q0;
        }
    }
    color = ((h.sample_class == SC_PLANAR16) | (h.sample_class == SC_PACKED16)) ? source_color_class16<false>(op, h, s, t, 0, 0)
            : h.sample_class != SC_GENERIC ? source_color_class<false>(op, h, s, t, 0, 0)
                                           : source_color_norm(op, s, t);
    return true;
}

/* ---- surfaces handed from one launch of a frame to the next (KSpill): UNORM codes <-> register values ---- */
template <int BITS>
DEV f4 load_px(const void* buf, size_t idx)
{
    if (BITS == 8) {
        const uint32_t v  = ((const uint32_t*)buf)[idx];
        const float    sc = 1.0f / 255.0f; /* == what quantize<8>() leaves in the register for code k */
        return f4{(float)(v & 0xff) * sc, (float)((v >> 8) & 0xff) * sc, (float)((v >> 16) & 0xff) * sc, (float)(v >> 24) * sc};
    }
    const uint2 v  = ((const uint2*)buf)[idx];
    const float sc = 1.0f / 65535.0f;
    return f4{(float)(v.x & 0xffff) * sc, (float)(v.x >> 16) * sc, (float)(v.y & 0xffff) * sc, (float)(v.y >> 16) * sc};
}
template <int BITS>
DEV void store_px(void* buf, size_t idx, f4 s)
{
    if (BITS == 8)
        ((uint32_t*)buf)[idx] = to_code<8>(s.x) | (to_code<8>(s.y) << 8) | (to_code<8>(s.z) << 16) | (to_code<8>(s.w) << 24);
    else
        ((uint2*)buf)[idx] = uint2{to_code<16>(s.x) | (to_code<16>(s.y) << 16), to_code<16>(s.z) | (to_code<16>(s.w) << 16)};
}
template <int BITS>
DEV float load_key(const void* buf, size_t idx)
{
    return BITS == 8 ? (float)((const uint8_t*)buf)[idx] * (1.0f / 255.0f) : (float)((const uint16_t*)buf)[idx] * (1.0f / 65535.0f);
}
template <int BITS>
DEV void store_key(void* buf, size_t idx, float k)
{
    if (BITS == 8)
        ((uint8_t*)buf)[idx] = (uint8_t)to_code<8>(k);
    else
        ((uint16_t*)buf)[idx] = (uint16_t)to_code<16>(k);
}

/* Which tile of the frame this workgroup composites. Workgroups are dealt round-robin over the 8 XCDs in dispatch
 * order (blocks b and b + 8 share one, MI355X_MICROARCH.md), each XCD with its own L2. Numbering tiles in that order
 * puts horizontally adjacent tiles on different XCDs, and every cache line two tiles share (the 4:2:x neighbour
 * texels, the halves of a 128-byte line) is then fetched from HBM once per XCD: the headline frame read 1.35 times
 * its source bytes. Large launches (sp.nbx != 0) are therefore one-dimensional and hand out tiles in chunks of
 * XCD_ROWS whole rows of tiles: chunk c goes to XCD c % 8, which walks it in row-major order (FETCH_SIZE 357 -> 243 MB
 * per launch, time unchanged: the kernel is bound by the vector ALU). Measured alternatives: one contiguous band of
 * the frame per XCD fetches the same and is 1 % faster on full-screen stacks, but draws that cover part of the frame
 * then load the XCDs unevenly (config 2: +14 %); a (8 * tiles per row, XCD_ROWS, chunks) grid needs no divisions but
 * ran 7 % slower. Small launches keep the plain two-dimensional numbering: their few MB sit in the Infinity Cache and
 * the renumbering's two scalar divisions and padding workgroups showed (config 1: +7 %), and so does the general
 * kernel, whose draws rarely cover whole rows. Pure renumbering: placement is never relied on. */
constexpr unsigned XCD_ROWS = 4;
DEV bool xcd_block(const KSpill& sp, unsigned& bx, unsigned& by)
{
    if (sp.nbx == 0) {
        bx = blockIdx.x, by = blockIdx.y;
        return true;
    }
    const unsigned nb = (unsigned)sp.nbx * (unsigned)sp.nby, C = (unsigned)sp.nbx * XCD_ROWS;
    const unsigned L = blockIdx.x, j = L >> 3; /* j: position in this XCD's sequence */
    const unsigned c = j / C, within = j - c * C;
    const unsigned id = (c * 8u + (L & 7u)) * C + within;
    if (id >= nb)
        return false;
    by = id / (unsigned)sp.nbx;
    bx = id - by * (unsigned)sp.nbx;
    return true;
}

/* the ops of chunk `chunk` (64 ops) that can touch the wave's band of the frame, as a bit mask (KSpill::masks) */
DEV uint64_t band_ops(const KSpill& sp, int chunk, int mstride, int mcol, int mrow, int nops)
{
    if (sp.dense) {
        const int n = nops - 64 * chunk;
        return n >= 64 ? ~0ull : ((1ull << (n > 0 ? n : 0)) - 1ull);
    }
    return sp.masks[chunk * mstride + mcol] & sp.masks[chunk * mstride + sp.ncols + mrow];
}

/* General compositor: one output pixel per lane (one wave = 64 consecutive pixels of a row, one
 * workgroup = 4 rows). FULL = false drops the two extra RGBA surfaces (layer / mix temporaries) from the
 * register file: the host picks it when the frame has only normal-blend, non-mix layers. Surfaces are
 * plain scalars selected by value, never arrays, so they stay in VGPRs. */
template <int BITS, bool FULL, bool FUSEDK = false> /* FUSEDK: the launch holds fused blend-mode layers (KOp::fused); an
                                                        instance of its own, because with that path compiled in,
                                                        launches without such ops ran 3 - 7 % slower */
__global__ __launch_bounds__(256) void cghip_composite(const KOp* __restrict__ ops, int nops, void* out, int W, int H,
                                                       int field_mode, KSpill sp)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned  bx, by;
    if (!xcd_block(sp, bx, by))
        return;
    const int y = __builtin_amdgcn_readfirstlane(sp.gy0 + (int)by * 4 + wave); /* wave-uniform: keep it scalar */
    if (y >= H)
        return;
    const int x = sp.gx0 + (int)bx * 64 + lane;

    f4    s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    float k0 = 0.0f, k1 = 0.0f;
    if (sp.load_mask != 0 && x < W) { /* surfaces an earlier launch of this frame left in HBM */
        const size_t idx = (size_t)y * W + x;
        if (sp.load_mask & (1u << KS_RGBA0)) s0 = load_px<BITS>(out, idx);
        if (FULL && (sp.load_mask & (1u << KS_RGBA1))) s1 = load_px<BITS>(sp.rgba1, idx);
        if (FULL && (sp.load_mask & (1u << KS_RGBA2))) s2 = load_px<BITS>(sp.rgba2, idx);
        if (sp.load_mask & (1u << KS_KEY0)) k0 = load_key<BITS>(sp.key0, idx);
        if (sp.load_mask & (1u << KS_KEY1)) k1 = load_key<BITS>(sp.key1, idx);
    }

    const bool row_live = !((field_mode == 1 && (y & 1) != 0) || (field_mode == 2 && (y & 1) != 1));

    /* only the ops that can touch this wave's band of the frame (KSpill::masks, host: op_masks) */
    const int       mcol = __builtin_amdgcn_readfirstlane(x) >> sp.col_shift, mrow = y >> 2;
    const int       mstride = sp.ncols + sp.nrows;
    for (int chunk = 0; chunk < sp.nchunks && row_live; ++chunk)
    for (uint64_t todo = band_ops(sp, chunk, mstride, mcol, mrow, nops); todo != 0; todo &= todo - 1) {
        const int  n   = chunk * 64 + __builtin_ctzll(todo);
        const KOp& op  = ops[__builtin_amdgcn_readfirstlane(n)]; /* n is uniform; say so (see dispatch_mode) */
        const HotG h   = load_hot_g<FUSEDK>(op);
        const bool clr = h.kind == KOP_CLEAR;
        {
            /* CLEAR as selects and ONE skip test: every early `continue` made the compiler copy all loop-carried
               surface registers at that exit (see cghip_composite_fast) */
            const bool c1 = h.dst == KS_RGBA1, c2 = h.dst == KS_RGBA2, c3 = h.dst == KS_KEY0, c4 = h.dst == KS_KEY1;
            if (clr) { /* wave-uniform; draws do not pay for the selects */
                if (FULL) {
                    s1 = sel4(c1, f4{0, 0, 0, 0}, s1);
                    s2 = sel4(c2, f4{0, 0, 0, 0}, s2);
                }
                k0 = c3 ? 0.0f : k0;
                k1 = c4 ? 0.0f : k1;
            }
        }
        /* cull against the op's touched box, then against the quad itself (a fused op: the composite's box culls, the
           item's own box and quad only decide whether its colour is zero) */
        bool live = !clr & (y >= h.y0) & (y < h.y1) & (x >= h.x0) & (x < h.x1);
        f4   color = {0, 0, 0, 0};
        bool covered = live;
        if (FUSEDK && h.fused) /* wave-uniform */
            covered = live & (y >= op.ay0) & (y < op.ay1) & (x >= op.ax0) & (x < op.ax1);
        if (covered) {
            if (h.kind == KOP_DRAW_SURFACE)
                color = FULL ? sel4(h.src_surface == KS_RGBA2, s2, s1) : f4{0, 0, 0, 0};
            else
                covered = source_color(op, h, x, y, color);
        }
        if (!(FUSEDK && h.fused))
            live = covered;
        if (!live)
            continue;

        const float lkey = h.local_key == KS_KEY1 ? k1 : k0;
        const float ykey = h.layer_key == KS_KEY1 ? k1 : k0;
        if (h.dst >= KS_KEY0) {
            /* R8/R16 target (image_mixer.cpp:174-183): background.bgra of a 1-channel texture is
             * (0,0,r,1); only the red slot of the result is stored. */
            const bool  one = h.dst == KS_KEY1;
            const f4    res = shade(op, color, lkey, ykey, f4{0.0f, 0.0f, one ? k1 : k0, 1.0f});
            const float q   = quantize<BITS>(res.z);
            k1              = one ? q : k1;
            k0              = one ? k0 : q;
        } else {
            const bool d1 = FULL && h.dst == KS_RGBA1, d2 = FULL && h.dst == KS_RGBA2;
            f4         back = s0;
            if (FULL)
                back = sel4(d1, s1, sel4(d2, s2, s0));
            /* ONE instance of the blend code: a fused op first runs the item over the cleared layer texture (normal
               blend over zero, no mode: a few multiplies), takes it through the texture's UNORM store, and then enters
               the common blend as the composite, with the composite's keys, opacity, keyer and mode */
            f4        src = shade_adjust(op, color);
            BlendArgs a   = {h.flags, h.local_key, h.layer_key, op.opacity, op.blend_mode};
            float     lk = lkey, yk = ykey;
            if (FUSEDK && h.fused) { /* wave-uniform */
                f4 t = shade_blend<MODE_NONE>(a, src, lkey, ykey, f4{0.0f, 0.0f, 0.0f, 0.0f});
                t    = f4{quantize<BITS>(t.x), quantize<BITS>(t.y), quantize<BITS>(t.z), quantize<BITS>(t.w)};
                src  = sel4(covered, t, f4{0.0f, 0.0f, 0.0f, 0.0f}); /* outside the item the texture is still zero */
                a    = BlendArgs{op.b_flags, op.b_local_key, op.b_layer_key, op.b_opacity, op.b_blend_mode};
                lk = a.local_key == KS_KEY1 ? k1 : k0, yk = a.layer_key == KS_KEY1 ? k1 : k0;
            }
            f4 res = shade_blend<MODE_RUNTIME>(a, src, lk, yk, back);
            res    = f4{quantize<BITS>(res.x), quantize<BITS>(res.y), quantize<BITS>(res.z), quantize<BITS>(res.w)};
            if (FULL) {
                s1 = sel4(d1, res, s1);
                s2 = sel4(d2, res, s2);
                s0 = sel4(!d1 && !d2, res, s0);
            } else {
                s0 = res;
            }
        }
    }

    /* frame store: memory order B,G,R,A = internal x,y,z,w (texture.cpp:100-105 reads back GL_BGRA) */
    if (x >= W)
        return;
    const size_t idx = (size_t)y * W + x;
    if (sp.store_mask & (1u << KS_RGBA0)) store_px<BITS>(out, idx, s0);
    if (FULL && (sp.store_mask & (1u << KS_RGBA1))) store_px<BITS>(sp.rgba1, idx, s1);
    if (FULL && (sp.store_mask & (1u << KS_RGBA2))) store_px<BITS>(sp.rgba2, idx, s2);
    if (sp.store_mask & (1u << KS_KEY0)) store_key<BITS>(sp.key0_st, idx, k0);
    if (sp.store_mask & (1u << KS_KEY1)) store_key<BITS>(sp.key1_st, idx, k1);
}

/* Adjustment parameters copied into VGPRs once per op, for the 4-pixels-per-lane kernel: a VALU instruction with an
 * SGPR operand issues at half rate on gfx950 (tests/ubench/valu_rate: v_mul_f32 v, s, v 1.8 ns vs 1.0 ns), and each
 * of these is used 4..16 times per lane and op. Same field names as KOp, so the shader functions take either. */
/* a scalar the compiler may not look through: arithmetic that depends on it stays where it is written */
DEV int opaque_s(int v)
{
    asm volatile("" : "+s"(v));
    return v;
}
DEV int opaque_v(int v) /* the same for a per-lane value */
{
    asm volatile("" : "+v"(v));
    return v;
}
DEV float in_vgpr(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}
struct LevelsV {
    float lv_min_input, lv_range, lv_inv_range, lv_inv_gamma, lv_min_output, lv_max_output;
    explicit DEV LevelsV(const KOp& o)
    { /* one batch of scalar loads and one wait (see ycoef16); gamma and the output range feed wave-uniform tests
         first: they stay scalar until used */
        float a = o.lv_min_input, b = o.lv_range, c = o.lv_inv_range, d = o.lv_inv_gamma, e = o.lv_min_output,
              f = o.lv_max_output;
        asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f));
        lv_min_input = in_vgpr(a), lv_range = in_vgpr(b), lv_inv_range = in_vgpr(c);
        lv_inv_gamma = d, lv_min_output = e, lv_max_output = f;
    }
};
struct ChromaV {
    float    ch_target_hue, ch_hue_width, ch_min_saturation, ch_min_brightness, ch_spill, ch_inv_spill,
        ch_spill_saturation, ch_soft_range, ch_inv_soft;
    uint32_t flags;
    explicit DEV ChromaV(const KOp& o)
        : flags(o.flags)
    { /* one batch of scalar loads and one wait (see ycoef16) */
        float a = o.ch_target_hue, b = o.ch_hue_width, c = o.ch_min_saturation, d = o.ch_min_brightness, e = o.ch_spill,
              f = o.ch_inv_spill, g = o.ch_spill_saturation, h = o.ch_soft_range, i = o.ch_inv_soft;
        asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h), "+s"(i));
        ch_target_hue = in_vgpr(a), ch_hue_width = in_vgpr(b), ch_min_saturation = in_vgpr(c);
        ch_min_brightness = in_vgpr(d), ch_spill = in_vgpr(e), ch_inv_spill = in_vgpr(f);
        ch_spill_saturation = in_vgpr(g), ch_soft_range = in_vgpr(h), ch_inv_soft = in_vgpr(i);
    }
};
struct CsbV {
    float    brt, sat, con;
    uint32_t flags;
    explicit DEV CsbV(const KOp& o)
        : flags(o.flags)
    {
        float a = o.brt, b = o.sat, c = o.con;
        asm volatile("" : "+s"(a), "+s"(b), "+s"(c));
        brt = in_vgpr(a), sat = in_vgpr(b), con = in_vgpr(c);
    }
};

/* ------------------------------------------------------------------ the fast compositor */

/* Fast compositor: frames whose every draw is a unit-scale, normal-blend layer whose pixel box and source offset
 * are multiples of 4 pixels (full-screen video stacks with key layers, chroma key, levels, CSB: BASELINE configs
 * 1, 3 and most of a live rundown). Four pixels per lane: a wave owns 256 consecutive pixels of a row, sources
 * arrive through aligned 4..32-byte loads, the frame leaves through 16/32-byte stores; a lane is either wholly
 * inside an op's box or wholly outside it, so there is no per-pixel predication. Numerics are those of
 * cghip_composite.
 *
 * The kernel is VALU-issue bound, so the per-op code is organised as a few STRAIGHT-LINE bodies picked by
 * wave-uniform tests (source layout x bit depth, then adjustments, then the blend variant) instead of per-pixel
 * switches: every wave-uniform branch inside a per-pixel function became a thicket of blocks, selects and
 * register copies. */

DEV float lerpf(float a, float b, float w) { return a + w * (b - a); }
DEV int   lerp8i(int a, int b, int w) { return a + ((w * (b - a) + 128) >> 8); }

/* The fields of an op that every fast-path draw needs, fetched together at the top of the op loop. Left to
 * itself the compiler loads each KOp field in the block that first uses it, and every such block then starts
 * with its own scalar-cache round trip (s_load + s_waitcnt): about ten serial stalls per op and wave. The empty
 * asm pins the values in SGPRs right here, so all of the loads are issued back to back and waited for once.
 * Adjustment parameters (levels, CSB, chroma key) stay lazily loaded inside their wave-uniform branches. */
struct HotOp {
    int32_t  kind;
    uint32_t flags;
    int32_t  dst, local_key, layer_key, pixel_format;
    int32_t  x0, y0, x1, y1, unit_dx, unit_dy;
    float    opacity;
    int32_t  fetch_body, blend_body;
    uint64_t ptr0, ptr1, ptr2, ptr3;
    int32_t  w0, w1, w2, w3, h1;
    int32_t  ax0, ax1; /* the item's own columns (== x0, x1 unless KOp::fused): what KF_EDGE masks against */
};

DEV HotOp load_hot(const KOp& op)
{
    HotOp h = {op.kind, op.flags, op.dst, op.local_key, op.layer_key, op.pixel_format, op.x0, op.y0, op.x1, op.y1,
               op.unit_dx, op.unit_dy, op.opacity, op.fetch_body, op.blend_body,
               (uint64_t)op.planes[0].ptr, (uint64_t)op.planes[1].ptr, (uint64_t)op.planes[2].ptr,
               (uint64_t)op.planes[3].ptr, op.planes[0].w, op.planes[1].w, op.planes[2].w, op.planes[3].w,
               op.planes[1].h, op.ax0, op.ax1};
    asm volatile(""
                 : "+s"(h.kind), "+s"(h.flags), "+s"(h.dst), "+s"(h.local_key), "+s"(h.layer_key), "+s"(h.pixel_format),
                   "+s"(h.x0), "+s"(h.y0), "+s"(h.x1), "+s"(h.y1), "+s"(h.unit_dx), "+s"(h.unit_dy), "+s"(h.opacity),
                   "+s"(h.fetch_body), "+s"(h.blend_body), "+s"(h.ptr0), "+s"(h.ptr1),
                   "+s"(h.ptr2), "+s"(h.ptr3), "+s"(h.w0), "+s"(h.w1), "+s"(h.w2), "+s"(h.w3), "+s"(h.h1),
                   "+s"(h.ax0), "+s"(h.ax1));
    return h;
}

/* raw texels of one plane row for a lane's four pixels: a, b = the row's dwords; half-width planes: a = the
 * aligned texel pair, b / c = left / right neighbour texel (clamped to the edge) */
struct Row3 {
    uint32_t a, b, c;
};

/* Row address = wave-uniform base (the row index is scalar: every lane of a wave is on the same output row) plus a
 * 32-bit per-lane byte offset, so the loads take the SGPR-base + VGPR-offset form instead of a 64-bit multiply-add
 * per lane and load. */
/* wave-uniform address of row r of a plane with w texels of B bytes per row: planes are smaller than 4 GiB
 * (cghip_runtime.cpp: plane_new), so the offset is ONE 32-bit scalar multiply, not a 64-bit product of sign-extended
 * operands (ten scalar instructions per plane and op) */
DEV GB row_base(uint64_t ptr, int r, int w, uint32_t B)
{
    return (GB)ptr + (size_t)((uint32_t)r * ((uint32_t)w * B));
}
template <int SB, int HALF>
DEV Row3 load_row(uint64_t ptr, int w, int r, int sx)
{
    Row3 o = {0, 0, 0};
    constexpr uint32_t B = SB / 8;
    const GB row = row_base(ptr, r, w, B); /* scalar */
    if (!HALF) {
        if (SB == 16) {
            const u32x2 v = load_at<u32x2>(row, (uint32_t)sx * B);
            o.a = v.x, o.b = v.y;
        } else {
            o.a = load_at<uint32_t>(row, (uint32_t)sx);
        }
    } else {
        const int      k  = sx >> 1;
        const uint32_t kl = (uint32_t)imax(k - 1, 0), kr = (uint32_t)imin(k + 2, w - 1);
        if (SB == 16) {
            o.a = load_at<uint32_t>(row, (uint32_t)k * B);
            o.b = load_at<uint16_t>(row, kl * B);
            o.c = load_at<uint16_t>(row, kr * B);
        } else {
            o.a = load_at<uint16_t>(row, (uint32_t)k);
            o.b = load_at<uint8_t>(row, kl);
            o.c = load_at<uint8_t>(row, kr);
        }
    }
    return o;
}

/* Four horizontally adjacent samples of a 1-channel plane row for output texels sx..sx+3 (sx % 4 == 0): at
 * unit scale (HALF == 0) or from a half-width plane (HALF == 1: texel centres fall 1/4 and 3/4 between
 * neighbours, weights 0.75 / 0.25 exactly). 16-bit: floats as texture() returns them, fp32 lerp. 8-bit:
 * integer codes after the rounded 8-bit-weight lerp, as floats (normalised later, after the optional row lerp). */
template <int SB, int HALF>
DEV void convert_row(const Row3 r, float o[4])
{
    if (SB == 16) {
        const float sc = 1.0f / 65535.0f;
        if (!HALF) {
            o[0] = (float)(r.a & 0xffff) * sc, o[1] = (float)(r.a >> 16) * sc;
            o[2] = (float)(r.b & 0xffff) * sc, o[3] = (float)(r.b >> 16) * sc;
        } else {
            const float t0 = (float)(r.a & 0xffff) * sc, t1 = (float)(r.a >> 16) * sc;
            const float tl = (float)r.b * sc, tr = (float)r.c * sc;
            o[0] = lerpf(tl, t0, 0.75f), o[1] = lerpf(t0, t1, 0.25f);
            o[2] = lerpf(t0, t1, 0.75f), o[3] = lerpf(t1, tr, 0.25f);
        }
    } else {
        if (!HALF) {
            o[0] = (float)(r.a & 0xff), o[1] = (float)((r.a >> 8) & 0xff);
            o[2] = (float)((r.a >> 16) & 0xff), o[3] = (float)(r.a >> 24);
        } else {
            const int t0 = r.a & 0xff, t1 = (r.a >> 8) & 0xff, tl = (int)r.b, tr = (int)r.c;
            o[0] = (float)lerp8i(tl, t0, 192), o[1] = (float)lerp8i(t0, t1, 64);
            o[2] = (float)lerp8i(t0, t1, 192), o[3] = (float)lerp8i(t1, tr, 64);
        }
    }
}

/* chroma plane: one row, or two rows lerped vertically (4:2:0), x first then y like the sampler; normalised */
template <int SB, int HX, int HY>
DEV void convert_chroma(const Row3 t, const Row3 u, int sy, float o[4])
{
    float a[4];
    convert_row<SB, HX>(t, a);
    if (HY) {
        /* v = (sy + .5)/2 - .5: even rows sit 3/4 of the way from row sy/2-1 to sy/2, odd rows 1/4 from (sy-1)/2 */
        const bool odd = sy & 1;
        float      b[4];
        convert_row<SB, HX>(u, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (SB == 16)
                a[i] = lerpf(a[i], b[i], odd ? 0.25f : 0.75f);
            else
                a[i] = (float)lerp8i((int)a[i], (int)b[i], odd ? 64 : 192);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = SB == 8 ? a[i] * (1.0f / 255.0f) : a[i];
}

/* ---- 8-bit planes: the YCbCr matrix's per-plane terms from tables in LDS ------------------------------------------
 * A filtered 8-bit sample is an integer code 0..255 (the sampler's 8-bit-weight lerps round to codes), and the first
 * half of ycbcra_to_rgba() is a function of ONE code each: 1.164 * (Y*255 - 16), and the four products k * (C*255 - 128).
 * Every workgroup fills the 256 entries of each with the very expressions of ycbcra_to_rgba() (one code per thread,
 * the unfused multiplies and subtractions of this build: bit-identical by construction) and the fetch bodies look them
 * up: 13 vector instructions and 3 LDS reads per pixel instead of 27 instructions. 9 KB per workgroup. Only in the
 * bit-exact build for 8-bit frames; the 4-pixels-per-lane kernels' aligned bodies (planar, uyvy). */
#ifndef CGHIP_YUV_LUT
#define CGHIP_YUV_LUT (CGHIP_FRAME_BITS == 8)
#endif
struct __attribute__((aligned(8))) f2 {
    float x, y;
};
struct YuvLut {
    float yy[256];    /* 1.164 * (Y * 255 - 16) */
    f2    cr[2][256]; /* [hd]: (k_rr * cr, k_gr * cr), cr = Cr * 255 - 128 */
    f2    cb[2][256]; /* [hd]: (k_gb * cb, k_bb * cb) */
};
#if CGHIP_YUV_LUT
__shared__ YuvLut g_yuv_lut;
DEV void fill_yuv_lut() /* every thread of the workgroup, before the first return; ends with the barrier */
{
    for (int c = (int)threadIdx.x; c < 256; c += (int)blockDim.x) {
        const float v = (float)c * (1.0f / 255.0f); /* what convert_chroma() / fetch_planar() made of a code */
        g_yuv_lut.yy[c] = 1.164f * (v * 255.0f - 16.0f);
        const float ch = v * 255.0f - 128.0f;
        g_yuv_lut.cr[0][c] = f2{1.596f * ch, 0.813f * ch};
        g_yuv_lut.cr[1][c] = f2{1.793f * ch, 0.534f * ch};
        g_yuv_lut.cb[0][c] = f2{0.391f * ch, 2.018f * ch};
        g_yuv_lut.cb[1][c] = f2{0.213f * ch, 2.115f * ch};
    }
    __syncthreads();
}
#endif
/* ycbcra_to_rgba(Y / 255, Cb / 255, Cr / 255, A, hd) for integer codes, from the tables where the build has them */
DEV f4 ycbcr_codes_to_rgba(int Y, int Cb, int Cr, float A, int hd)
{
#if CGHIP_YUV_LUT
    const float yy = g_yuv_lut.yy[Y];
    const f2    a = g_yuv_lut.cr[hd][Cr], b = g_yuv_lut.cb[hd][Cb];
    f4          c;
    c.z = divc<RXF>((yy + a.x), 255.0f, 1.0f / 255.0f);
    c.y = divc<RXF>((yy - a.y - b.x), 255.0f, 1.0f / 255.0f);
    c.x = divc<RXF>((yy + b.y), 255.0f, 1.0f / 255.0f);
    c.w = A;
    return c;
#else
    const float sc = 1.0f / 255.0f;
    return ycbcra_to_rgba((float)Y * sc, (float)Cb * sc, (float)Cr * sc, A, hd != 0);
#endif
}
/* convert_row<8, HALF> / convert_chroma<8, HX, HY> with the codes left as integers */
template <int HALF>
DEV void codes_row8(const Row3 r, int o[4])
{
    if (!HALF) {
        o[0] = (int)(r.a & 0xff), o[1] = (int)((r.a >> 8) & 0xff), o[2] = (int)((r.a >> 16) & 0xff), o[3] = (int)(r.a >> 24);
    } else {
        const int t0 = r.a & 0xff, t1 = (r.a >> 8) & 0xff, tl = (int)r.b, tr = (int)r.c;
        o[0] = lerp8i(tl, t0, 192), o[1] = lerp8i(t0, t1, 64), o[2] = lerp8i(t0, t1, 192), o[3] = lerp8i(t1, tr, 64);
    }
}
template <int HX, int HY>
DEV void codes_chroma8(const Row3 t, const Row3 u, int sy, int o[4])
{
    codes_row8<HX>(t, o);
    if (HY) {
        const bool odd = sy & 1;
        int        b[4];
        codes_row8<HX>(u, b);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = lerp8i(o[i], b[i], odd ? 64 : 192);
    }
}

/* source colours of four pixels from ycbcr / ycbcra planes (shader.frag:441-467,500-512) */
template <int SB, int HX, int HY>
DEV void fetch_planar(const HotOp& op, int sx, int sy, bool red_only, f4 col[4])
{
    /* red_only (wave-uniform): the draw lands in a key surface and has no colour adjustment, so only the red and
       alpha slots are ever looked at (image_mixer.cpp:174-183): R = Y' + k*Cr' needs neither the Cb plane nor G, B */
    const int  r0    = (sy - 1) >> 1;
    const int  ra    = HY ? iclamp(r0, 0, op.h1 - 1) : sy;
    const int  rb    = HY ? iclamp(r0 + 1, 0, op.h1 - 1) : sy;
    const bool alpha = op.pixel_format == 6;
    /* all loads first: one HBM round trip per op, not one per plane */
    const Row3 y   = load_row<SB, 0>(op.ptr0, op.w0, sy, sx);
    const Row3 cr0 = load_row<SB, HX>(op.ptr2, op.w2, ra, sx);
    Row3       cb0 = {0, 0, 0}, cb1 = {0, 0, 0}, cr1 = {0, 0, 0}, al = {0, 0, 0};
    if (HY)
        cr1 = load_row<SB, HX>(op.ptr2, op.w2, rb, sx);
    if (!red_only) {
        cb0 = load_row<SB, HX>(op.ptr1, op.w1, ra, sx);
        if (HY)
            cb1 = load_row<SB, HX>(op.ptr1, op.w1, rb, sx);
    }
    if (alpha)
        al = load_row<SB, 0>(op.ptr3, op.w3, sy, sx);

    const float sc = SB == 8 ? 1.0f / 255.0f : 1.0f;
#if CGHIP_YUV_LUT
    if constexpr (SB == 8) {
        float Al[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (alpha) {
            convert_row<8, 0>(al, Al);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                Al[i] *= sc;
        }
        const int hd = (op.flags & KF_IS_HD) ? 1 : 0;
        int       Yc[4], Crc[4];
        float     yy[4];
        f2        a[4];
        codes_row8<0>(y, Yc);
        codes_chroma8<HX, HY>(cr0, cr1, sy, Crc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            yy[i]    = g_yuv_lut.yy[Yc[i]];
            a[i]     = g_yuv_lut.cr[hd][Crc[i]];
            col[i].z = divc<RXF>((yy[i] + a[i].x), 255.0f, 1.0f / 255.0f);
            col[i].w = Al[i];
            col[i].x = col[i].y = 0.0f;
        }
        if (!red_only) {
            int Cbc[4];
            codes_chroma8<HX, HY>(cb0, cb1, sy, Cbc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2 b = g_yuv_lut.cb[hd][Cbc[i]];
                col[i].y   = divc<RXF>((yy[i] - a[i].y - b.x), 255.0f, 1.0f / 255.0f);
                col[i].x   = divc<RXF>((yy[i] + b.y), 255.0f, 1.0f / 255.0f);
            }
        }
        return;
    }
#endif
    float       Y[4], Cr[4], Al[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    convert_row<SB, 0>(y, Y);
    convert_chroma<SB, HX, HY>(cr0, cr1, sy, Cr);
    if (alpha) {
        convert_row<SB, 0>(al, Al);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            Al[i] *= sc;
    }
    /* coefficients picked by scalar selects: same arithmetic as ycbcra_to_rgba() */
    const bool  hd  = op.flags & KF_IS_HD;
    const float krr = hd ? 1.793f : 1.596f, kgr = hd ? 0.534f : 0.813f, kgb = hd ? 0.213f : 0.391f,
                kbb = hd ? 2.115f : 2.018f;
    float yy[4], cr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        yy[i]    = 1.164f * ((SB == 8 ? Y[i] * sc : Y[i]) * 255.0f - 16.0f);
        cr[i]    = Cr[i] * 255.0f - 128.0f;
        col[i].z = divc<RXF>((yy[i] + krr * cr[i]), 255.0f, 1.0f / 255.0f);
        col[i].w = Al[i];
        col[i].x = col[i].y = 0.0f;
    }
    if (!red_only) {
        float Cb[4];
        convert_chroma<SB, HX, HY>(cb0, cb1, sy, Cb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float cb = Cb[i] * 255.0f - 128.0f;
            col[i].y       = divc<RXF>((yy[i] - kgr * cr[i] - kgb * cb), 255.0f, 1.0f / 255.0f);
            col[i].x       = divc<RXF>((yy[i] + kbb * cb), 255.0f, 1.0f / 255.0f);
        }
    }
}

/* source colours of four pixels from an 8-bit uyvy frame at unit scale (pixel format 10: what SDI capture delivers,
 * modules/ffmpeg/util/av_util.cpp:163-171). The frame is ONE buffer of U Y0 V Y1 groups bound twice: as an RG8 plane
 * w x h, whose .g is luma at the pixel's own texel, and as an RGBA8 plane (w/2) x h, whose .b / .r are Cb / Cr, filtered
 * across the half-width texels with the exact 1/4 - 3/4 weights like a 4:2:2 chroma plane (shader.frag:513-519). A
 * lane's four pixels are two such groups: one 8-byte load, plus the group to the left and to the right (clamped). */
DEV void fetch_uyvy8(const HotOp& op, int sx, int sy, f4 col[4])
{
    const GB       row = row_base(op.ptr1, sy, op.w1, 4u); /* scalar */
    const int      k   = sx >> 1;                          /* first of the lane's two groups */
    const u32x2    g   = load_at<u32x2>(row, (uint32_t)k * 4u);
    const uint32_t gl  = load_at<uint32_t>(row, (uint32_t)imax(k - 1, 0) * 4u);
    const uint32_t gr  = load_at<uint32_t>(row, (uint32_t)imin(k + 2, op.w1 - 1) * 4u);
#if CGHIP_YUV_LUT
    {
        const int Yc[4] = {(int)((g.x >> 8) & 0xff), (int)(g.x >> 24), (int)((g.y >> 8) & 0xff), (int)(g.y >> 24)};
        int       Cbc[4], Crc[4];
        codes_row8<1>(Row3{(g.x & 0xff) | ((g.y & 0xff) << 8), gl & 0xff, gr & 0xff}, Cbc);
        codes_row8<1>(Row3{((g.x >> 16) & 0xff) | (((g.y >> 16) & 0xff) << 8), (gl >> 16) & 0xff, (gr >> 16) & 0xff}, Crc);
        const int hd = (op.flags & KF_IS_HD) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { /* ycbcra_to_rgba() from the tables, as in fetch_planar */
            const float yy = g_yuv_lut.yy[Yc[i]];
            const f2    a = g_yuv_lut.cr[hd][Crc[i]], b = g_yuv_lut.cb[hd][Cbc[i]];
            col[i].z = divc<RXF>((yy + a.x), 255.0f, 1.0f / 255.0f);
            col[i].y = divc<RXF>((yy - a.y - b.x), 255.0f, 1.0f / 255.0f);
            col[i].x = divc<RXF>((yy + b.y), 255.0f, 1.0f / 255.0f);
            col[i].w = 1.0f;
        }
        return;
    }
#endif
    const float    sc  = 1.0f / 255.0f;
    float          Y[4] = {(float)((g.x >> 8) & 0xff), (float)(g.x >> 24), (float)((g.y >> 8) & 0xff), (float)(g.y >> 24)};
    float          Cb[4], Cr[4];
    convert_row<8, 1>(Row3{(g.x & 0xff) | ((g.y & 0xff) << 8), gl & 0xff, gr & 0xff}, Cb);
    convert_row<8, 1>(Row3{((g.x >> 16) & 0xff) | (((g.y >> 16) & 0xff) << 8), (gl >> 16) & 0xff, (gr >> 16) & 0xff}, Cr);
    const bool  hd  = op.flags & KF_IS_HD;
    const float krr = hd ? 1.793f : 1.596f, kgr = hd ? 0.534f : 0.813f, kgb = hd ? 0.213f : 0.391f,
                kbb = hd ? 2.115f : 2.018f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { /* ycbcra_to_rgba(), as in fetch_planar */
        const float yy = 1.164f * ((Y[i] * sc) * 255.0f - 16.0f);
        const float cr = (Cr[i] * sc) * 255.0f - 128.0f, cb = (Cb[i] * sc) * 255.0f - 128.0f;
        col[i].z = divc<RXF>((yy + krr * cr), 255.0f, 1.0f / 255.0f);
        col[i].y = divc<RXF>((yy - kgr * cr - kgb * cb), 255.0f, 1.0f / 255.0f);
        col[i].x = divc<RXF>((yy + kbb * cb), 255.0f, 1.0f / 255.0f);
        col[i].w = 1.0f;
    }
}

/* source colours of four pixels from one 8-bit plane of 1 channel (gray, luma) or 3 channels (bgr, rgb) at unit
 * scale: the plane's texels as GL returns them (gl_texel), then the format's swizzle (shader.frag:485,499-505) - the
 * two functions the general kernel's per-format path uses */
DEV void fetch_bytes8(const HotOp& op, int sx, int sy, f4 col[4])
{
    const float sc = 1.0f / 255.0f;
    if (op.pixel_format == 8 || op.pixel_format == 9) { /* wave-uniform: 12 bytes = four B,G,R texels */
        const GB       row = row_base(op.ptr0, sy, op.w0, 3u);
        const uint32_t a = load_at<uint32_t>(row, (uint32_t)sx * 3u), b = load_at<uint32_t>(row, (uint32_t)sx * 3u + 4u),
                       c = load_at<uint32_t>(row, (uint32_t)sx * 3u + 8u);
        const uint32_t t[4] = {a & 0xffffffu, (a >> 24) | ((b & 0xffffu) << 8), (b >> 16) | ((c & 0xffu) << 16), c >> 8};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            col[i] = swizzle_format(op.pixel_format, gl_texel(3, (float)(t[i] & 0xff) * sc, (float)((t[i] >> 8) & 0xff) * sc,
                                                              (float)((t[i] >> 16) & 0xff) * sc, 0.0f));
    } else {
        const uint32_t v = load_at<uint32_t>(row_base(op.ptr0, sy, op.w0, 1u), (uint32_t)sx);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            col[i] = swizzle_format(op.pixel_format, gl_texel(1, (float)((v >> (8 * i)) & 0xff) * sc, 0.0f, 0.0f, 0.0f));
    }
}

/* ---- the same for 16-bit planes in 16-bit frames (relaxed numerics, see divf) ----
 * Between the sample codes and the colour there is nothing but affine maps: code / 65535, the 1/4 - 3/4 lerps of the
 * half-resolution chroma planes, Y' = 1.164 (255 Y - 16), the matrix, / 255 (shader.frag:441-467). Composed:
 *     R = aY Yc + aR Crc + cR,  G = aY Yc + aGr Crc + aGb Cbc + cG,  B = aY Yc + aB Cbc + cB
 * on the codes Yc and the lerped chroma codes Cbc, Crc, with coefficients rounded once from double: 7 fmas per pixel
 * instead of 27 operations, a few 1e-8 from the exact value where the shader's own sequence is a few 1e-7 from it. */
struct YCoef {
    float aY, aR, cR, aGr, aGb, cG, aB, cB;
};
DEV YCoef ycoef16(const KOp& o) /* from the planner (cghip_plan.cpp: classify_fast), moved to VGPRs once per op */
{
    /* all eight pinned in SGPRs first, so that they arrive in ONE scalar load and one wait: with in_vgpr() straight on
       the fields every coefficient was a load, a wait and a move of its own, eight scalar-cache round trips in a row
       per op and wave */
    float k0 = o.ycoef[0], k1 = o.ycoef[1], k2 = o.ycoef[2], k3 = o.ycoef[3], k4 = o.ycoef[4], k5 = o.ycoef[5],
          k6 = o.ycoef[6], k7 = o.ycoef[7];
    asm volatile("" : "+s"(k0), "+s"(k1), "+s"(k2), "+s"(k3), "+s"(k4), "+s"(k5), "+s"(k6), "+s"(k7));
    return YCoef{in_vgpr(k0), in_vgpr(k1), in_vgpr(k2), in_vgpr(k3), in_vgpr(k4), in_vgpr(k5), in_vgpr(k6), in_vgpr(k7)};
}
/* four chroma CODES (not normalised) of one plane row for a lane's pixels */
template <int HX>
DEV void chroma_codes_row(const Row3 r, float o[4])
{
    if (!HX) {
        o[0] = (float)(r.a & 0xffff), o[1] = (float)(r.a >> 16), o[2] = (float)(r.b & 0xffff), o[3] = (float)(r.b >> 16);
    } else {
        const float t0 = (float)(r.a & 0xffff), t1 = (float)(r.a >> 16), tl = (float)r.b, tr = (float)r.c;
        const float d  = t1 - t0;
        o[0] = __builtin_fmaf(0.75f, t0 - tl, tl), o[1] = __builtin_fmaf(0.25f, d, t0);
        o[2] = __builtin_fmaf(0.75f, d, t0), o[3] = __builtin_fmaf(0.25f, tr - t1, t1);
    }
}
template <int HX, int HY>
DEV void chroma_codes(const Row3 t, const Row3 u, int sy, float o[4])
{
    chroma_codes_row<HX>(t, o);
    if (HY) {
        float b[4];
        chroma_codes_row<HX>(u, b);
        const float w = (sy & 1) ? 0.25f : 0.75f; /* see convert_chroma */
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = __builtin_fmaf(w, b[i] - o[i], o[i]);
    }
}
/* byte offsets of a lane's chroma texels in a half-width row: the aligned pair, the left and the right neighbour
 * (clamped to the edge); Cb and Cr planes have one geometry, so one set serves both */
struct ChromaOff {
    uint32_t a, l, r;
};
template <int HX>
DEV Row3 load_chroma16(uint64_t ptr, int w, int row, int sx, const ChromaOff o)
{
    Row3     v = {0, 0, 0};
    const GB b = row_base(ptr, row, w, 2u); /* scalar */
    if (!HX) {
        const u32x2 q = load_at<u32x2>(b, (uint32_t)sx * 2u);
        v.a = q.x, v.b = q.y;
    } else {
        v.a = load_at<uint32_t>(b, o.a);
        v.b = load_at<uint16_t>(b, o.l);
        v.c = load_at<uint16_t>(b, o.r);
    }
    return v;
}
/* a value nobody looks at: the compiler may leave whatever the register holds (no instruction) */
DEV float dont_care() { return __builtin_nondeterministic_value(0.0f); }
template <int HX, int HY>
DEV void fetch_planar16r(const KOp& kop, const HotOp& op, int sx, int sy, bool red_only, f4 col[4])
{
    const int  r0    = (sy - 1) >> 1;
    const int  ra    = HY ? iclamp(r0, 0, op.h1 - 1) : sy;
    const int  rb    = HY ? iclamp(r0 + 1, 0, op.h1 - 1) : sy;
    const bool alpha = op.pixel_format == 6;
    ChromaOff  co    = {0, 0, 0};
    if (HX) {
        const int k = sx >> 1;
        co.a = (uint32_t)k * 2u, co.l = (uint32_t)imax(k - 1, 0) * 2u, co.r = (uint32_t)imin(k + 2, op.w1 - 1) * 2u;
    }
    /* Two straight-line variants, picked by the wave-uniform red_only: a key draw without colour adjustments (only red
       and alpha are ever looked at, image_mixer.cpp:174-183: no Cb plane, no green, no blue) and the full colour. As ONE
       body with `if (!red_only)` around the Cb loads and the green / blue rows, every value those ifs define needed a
       default: 23 v_mov per op and lane in front of the branches, a tenth of a plain layer's instructions. Each
       variant issues all its loads first (the full one starts with the Cb rows, so that the two do not begin alike:
       the compiler hoists a common beginning in front of the branch, waits for those loads there, and the Cb rows
       became a second round trip). */
    if (red_only) {
        const u32x2 yv  = load_at<u32x2>(row_base(op.ptr0, sy, op.w0, 2u), (uint32_t)sx * 2u);
        const Row3  cr0 = load_chroma16<HX>(op.ptr2, op.w1, ra, sx, co);
        Row3        cr1 = {0, 0, 0};
        if (HY)
            cr1 = load_chroma16<HX>(op.ptr2, op.w1, rb, sx, co);
        const YCoef k = ycoef16(kop);
        const float Y[4] = {(float)(yv.x & 0xffff), (float)(yv.x >> 16), (float)(yv.y & 0xffff), (float)(yv.y >> 16)};
        float       Cr[4];
        chroma_codes<HX, HY>(cr0, cr1, sy, Cr);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            col[i].z = __builtin_fmaf(Y[i], k.aY, __builtin_fmaf(Cr[i], k.aR, k.cR));
            col[i].w = 1.0f;
            col[i].x = col[i].y = dont_care();
        }
    } else {
        const Row3 cb0 = load_chroma16<HX>(op.ptr1, op.w1, ra, sx, co);
        Row3       cb1 = {0, 0, 0}, cr1 = {0, 0, 0};
        if (HY)
            cb1 = load_chroma16<HX>(op.ptr1, op.w1, rb, sx, co);
        const Row3  cr0 = load_chroma16<HX>(op.ptr2, op.w1, ra, sx, co);
        if (HY)
            cr1 = load_chroma16<HX>(op.ptr2, op.w1, rb, sx, co);
        const u32x2 yv  = load_at<u32x2>(row_base(op.ptr0, sy, op.w0, 2u), (uint32_t)sx * 2u);
        const YCoef k = ycoef16(kop);
        const float Y[4] = {(float)(yv.x & 0xffff), (float)(yv.x >> 16), (float)(yv.y & 0xffff), (float)(yv.y >> 16)};
        float       Cr[4], Cb[4];
        chroma_codes<HX, HY>(cr0, cr1, sy, Cr);
        chroma_codes<HX, HY>(cb0, cb1, sy, Cb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            col[i].z = __builtin_fmaf(Y[i], k.aY, __builtin_fmaf(Cr[i], k.aR, k.cR));
            col[i].y = __builtin_fmaf(Y[i], k.aY, __builtin_fmaf(Cr[i], k.aGr, __builtin_fmaf(Cb[i], k.aGb, k.cG)));
            col[i].x = __builtin_fmaf(Y[i], k.aY, __builtin_fmaf(Cb[i], k.aB, k.cB));
            col[i].w = 1.0f;
        }
    }
    if (alpha) { /* wave-uniform and rare (ycbcra): a round trip of its own; the empty asm keeps it a branch */
        asm volatile("");
        const u32x2 av = load_at<u32x2>(row_base(op.ptr3, sy, op.w0, 2u), (uint32_t)sx * 2u);
        col[0].w = (float)(av.x & 0xffff) * (1.0f / 65535.0f), col[1].w = (float)(av.x >> 16) * (1.0f / 65535.0f);
        col[2].w = (float)(av.y & 0xffff) * (1.0f / 65535.0f), col[3].w = (float)(av.y >> 16) * (1.0f / 65535.0f);
    }
}

/* source colours of four pixels from one 4-channel plane (bgra / rgba / argb / abgr, shader.frag:487-498). The
 * format swizzle is a byte permute of the raw texel (selector picked by scalar selects), so the conversion that
 * follows is the same straight line for all four formats. Internal order x,y,z,w = memory samples
 * (0,1,2,3) for bgra, (2,1,0,3) rgba, (3,2,1,0) argb, (1,0,3,2) abgr. */
template <int SB>
DEV void fetch_packed(const HotOp& op, int sx, int sy, f4 col[4])
{
    const int pf = op.pixel_format;
    if (SB == 8) {
        const uint32_t sel  = pf == 1 ? 0x03020100u : pf == 2 ? 0x03000102u : pf == 3 ? 0x00010203u : 0x02030001u;
        u32x4          v;
        if (op.flags & KF_SOLID) { /* one texel for every pixel (wave-uniform branch) */
            const uint32_t t = load_at<uint32_t>((GB)op.ptr0, 0u);
            v                = u32x4{t, t, t, t};
        } else {
            v = load_at<u32x4>(row_base(op.ptr0, sy, op.w0, 4u), (uint32_t)sx * 4u);
        }
        const uint32_t m[4] = {v.x, v.y, v.z, v.w};
        const float    sc   = 1.0f / 255.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t = __builtin_amdgcn_perm(m[i], m[i], sel);
            col[i] = f4{(float)(t & 0xff) * sc, (float)((t >> 8) & 0xff) * sc, (float)((t >> 16) & 0xff) * sc,
                        (float)(t >> 24) * sc};
        }
    } else {
        const uint32_t slo = pf == 1 ? 0x03020100u : pf == 2 ? 0x03020504u : pf == 3 ? 0x05040706u : 0x01000302u;
        const uint32_t shi = pf == 1 ? 0x07060504u : pf == 2 ? 0x07060100u : pf == 3 ? 0x01000302u : 0x05040706u;
        u32x4          v, w;
        if (op.flags & KF_SOLID) {
            const u32x2 t = load_at<u32x2>((GB)op.ptr0, 0u);
            v = w = u32x4{t.x, t.y, t.x, t.y};
        } else {
            const GB row = row_base(op.ptr0, sy, op.w0, 8u);
            v = load_at<u32x4>(row, (uint32_t)sx * 8u), w = load_at<u32x4>(row, (uint32_t)sx * 8u + 16u);
        }
        const uint32_t m[8] = {v.x, v.y, v.z, v.w, w.x, w.y, w.z, w.w};
        const float    sc   = 1.0f / 65535.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = __builtin_amdgcn_perm(m[2 * i + 1], m[2 * i], slo);
            const uint32_t hi = __builtin_amdgcn_perm(m[2 * i + 1], m[2 * i], shi);
            col[i] = f4{(float)(lo & 0xffff) * sc, (float)(lo >> 16) * sc, (float)(hi & 0xffff) * sc,
                        (float)(hi >> 16) * sc};
        }
    }
}

/* shader.frag:532-541 for four pixels in two steps. (1) fore4 / fore_key4: the fore colour after keys, opacity and
 * invert, and the factor `ia` the background is multiplied with (1 - fore.a, or 1 for the additive keyer), in the
 * variant the planner picked (KOp::blend_body); the variants drop the neutral-operand steps of shade_blend() (x * 1,
 * fma(x, 1, 0) are exact, so the results are the same). (2) ONE place where the surface registers are rewritten,
 * store_rgba4 / the key select: when every variant wrote the loop-carried surfaces itself, the variants' results met at
 * the loop latch in different registers and every op paid 26 register copies there. */
struct Fore {
    f4    f[4];
    float ia[4];
};
/* VARIANT 0: no keys, no invert, linear keyer. 1: exactly one key (local or layer), else like 0: the usual keyed
 * fill. 2: everything, optional steps made unconditional with neutral operands. */
template <int VARIANT>
DEV void fore4(const HotOp& op, const f4 col[4], const float K0[4], const float K1[4], Fore& o)
{
    float opa = op.opacity;
    asm volatile("" : "+v"(opa)); /* VALU ops with an SGPR operand issue at ~2/3 rate (tests/ubench): one v_mov instead */
    if (VARIANT == 0) {
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            o.f[i]  = f4{col[i].x * opa, col[i].y * opa, col[i].z * opa, col[i].w * opa};
            o.ia[i] = 1.0f - o.f[i].w;
        });
    } else if (VARIANT == 1) {
        const bool from1 = (op.local_key != KS_NONE ? op.local_key : op.layer_key) == KS_KEY1;
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const float   k = from1 ? K1[i] : K0[i];
            f4            c = col[i];
            c.x *= k, c.y *= k, c.z *= k, c.w *= k;
            o.f[i]  = f4{c.x * opa, c.y * opa, c.z * opa, c.w * opa};
            o.ia[i] = 1.0f - o.f[i].w;
        });
    } else {
        const bool  lk = op.local_key != KS_NONE, yk = op.layer_key != KS_NONE;
        const bool  l1 = op.local_key == KS_KEY1, y1 = op.layer_key == KS_KEY1;
        const bool  inv = op.flags & KF_INVERT, add = op.flags & KF_ADDITIVE;
        const float im = inv ? -1.0f : 1.0f, ia0 = inv ? 1.0f : 0.0f;
        static_for<0, 4>([&](auto I) {
            constexpr int i  = decltype(I)::value;
            const float   k1 = lk ? (l1 ? K1[i] : K0[i]) : 1.0f;
            const float   k2 = yk ? (y1 ? K1[i] : K0[i]) : 1.0f;
            f4            c  = col[i];
            c.x *= k1, c.y *= k1, c.z *= k1, c.w *= k1;
            c.x *= k2, c.y *= k2, c.z *= k2, c.w *= k2;
            c.x *= opa, c.y *= opa, c.z *= opa, c.w *= opa;
            o.f[i]  = f4{__builtin_fmaf(c.x, im, ia0), __builtin_fmaf(c.y, im, ia0), __builtin_fmaf(c.z, im, ia0),
                         __builtin_fmaf(c.w, im, ia0)};
            o.ia[i] = add ? 1.0f : 1.0f - o.f[i].w;
        });
    }
}
/* the same for a draw into a key surface: only the red slot is stored (image_mixer.cpp:174-183), so only red (f.z)
 * and alpha (f.w) are evaluated. Keys of a key draw, if any, come from KEY0 like in cghip_composite. */
template <bool SIMPLE>
DEV void fore_key4(const HotOp& op, const f4 col[4], const float K0[4], Fore& o)
{
    float opa = op.opacity;
    asm volatile("" : "+v"(opa));
    const bool  lk = op.local_key != KS_NONE, yk = op.layer_key != KS_NONE;
    const bool  inv = op.flags & KF_INVERT, add = op.flags & KF_ADDITIVE;
    const float im = inv ? -1.0f : 1.0f, ia0 = inv ? 1.0f : 0.0f;
    static_for<0, 4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        float         z = col[i].z, w = col[i].w, ia;
        if (SIMPLE) {
            z *= opa, w *= opa;
            ia = 1.0f - w;
        } else {
            const float k1 = lk ? K0[i] : 1.0f, k2 = yk ? K0[i] : 1.0f;
            z *= k1, w *= k1;
            z *= k2, w *= k2;
            z *= opa, w *= opa;
            z  = __builtin_fmaf(z, im, ia0), w = __builtin_fmaf(w, im, ia0);
            ia = add ? 1.0f : 1.0f - w;
        }
        o.f[i]  = f4{0.0f, 0.0f, z, w};
        o.ia[i] = ia;
    });
}
/* fore + ia * back, then the UNORM store (shader.frag:429-433,541) */
#ifndef CGHIP_INPLACE_STORE
#define CGHIP_INPLACE_STORE 0
#endif
/* The relaxed build's three fmas of quantize<16>(f + ia * s) (clamp as output modifier, magic-constant rounding, back to
 * [0,1]) written so that the surface register is updated WHERE IT LIVES: left to the compiler the sixteen results of
 * an op landed in whatever registers the source colour had freed and were copied into the loop-carried surface
 * registers at the loop latch (16 v_mov per op and lane, 8 % of a plain layer's instructions). Tied operands leave it no
 * choice. Same instructions, same constants (0x3f7fff00 = 65535/65536, 0x3f800080 = 65536/65535). */
DEV void blend_store16_inplace(float& s, float ia, float f, float magic, float neg_md)
{
    asm("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(s) : "v"(ia), "v"(f));
    asm("v_fmamk_f32 %0, %0, 0x3f7fff00, %1" : "+v"(s) : "v"(magic));
    asm("v_fmamk_f32 %0, %0, 0x3f800080, %1" : "+v"(s) : "v"(neg_md));
}
template <int BITS>
DEV void store_rgba4(const Fore& o, f4 S[4])
{
    if constexpr (RXF && BITS == 16 && CGHIP_INPLACE_STORE) {
        const float magic = in_vgpr(Q<16>::magic), neg_md = in_vgpr(-Q<16>::magic * Q<16>::dequant);
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            blend_store16_inplace(S[i].x, o.ia[i], o.f[i].x, magic, neg_md);
            blend_store16_inplace(S[i].y, o.ia[i], o.f[i].y, magic, neg_md);
            blend_store16_inplace(S[i].z, o.ia[i], o.f[i].z, magic, neg_md);
            blend_store16_inplace(S[i].w, o.ia[i], o.f[i].w, magic, neg_md);
        });
        return;
    }
    static_for<0, 4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        S[i] = f4{quantize<BITS>(o.f[i].x + o.ia[i] * S[i].x), quantize<BITS>(o.f[i].y + o.ia[i] * S[i].y),
                  quantize<BITS>(o.f[i].z + o.ia[i] * S[i].z), quantize<BITS>(o.f[i].w + o.ia[i] * S[i].w)};
    });
}
template <int BITS>
DEV void store_key4(const Fore& o, bool key1, float K0[4], float K1[4])
{
    static_for<0, 4>([&](auto I) {
        constexpr int i  = decltype(I)::value;
        const float   a0 = K0[i], a1 = K1[i];
        const float   q  = quantize<BITS>(o.f[i].z + o.ia[i] * (key1 ? a1 : a0));
        K0[i]            = key1 ? a0 : q;
        K1[i]            = key1 ? q : a1;
    });
}

/* ---- sampled sources in the 4-pixels-per-lane kernel ----
 * FB_RECT8: an axis-aligned scaled rectangle (picture-in-picture boxes: image_kernel.cpp:195-205 with angle 0) of 8-bit
 * ycbcr(a) planes or one packed 4-channel plane. KF_AXIS says that s depends on x alone and t on y alone (the other
 * gradients are exactly zero), so everything that hangs on t - the two source rows, their clamps, the row weight - is
 * the same for the whole wave and lives in scalar registers; a pixel computes its column taps and twelve (four) loads
 * through scalar row bases. Same arithmetic as source_color + source_color_class<false>, term for term: fma(0, x, c)
 * is c, and (v * size) * 256 equals v * (size * 256) because scaling by 2^8 commutes with rounding. */
struct RowTaps { /* wave-uniform */
    GB  r0, r1; /* row y0 and row y1 of the plane */
    int wy;
};
DEV RowTaps row_taps(uint64_t ptr, int iv, int w, int h, uint32_t texel_bytes)
{
    int       y0 = iv >> 8;
    const int y1 = iclamp(y0 + 1, 0, h - 1);
    y0           = iclamp(y0, 0, h - 1);
    const uint32_t pitch = (uint32_t)w * texel_bytes;
    return RowTaps{(GB)ptr + (size_t)((uint32_t)y0 * pitch), (GB)ptr + (size_t)((uint32_t)y1 * pitch), iv & 0xff};
}
/* Per pixel: the two column taps are neighbours, so ONE load of two texels at column `x` serves both (planes are at
 * least two texels wide, cghip_plan.cpp). CLAMP_TO_EDGE: left of the plane both taps are texel 0 - weight 0 on the pair
 * (0, 1); right of column w - 2 both are texel w - 1 - weight 256 on the pair (w - 2, w - 1), and lerp8 with weight 256
 * returns its second operand exactly. */
struct ColTaps {
    uint32_t x;
    int      wx;
};
DEV ColTaps col_taps(int iu, int w)
{
    const int x0 = iu >> 8;
    int       wx = iu & 0xff;
    wx           = x0 < 0 ? 0 : wx;
    wx           = x0 > w - 2 ? 256 : wx;
    return ColTaps{(uint32_t)iclamp(x0, 0, w - 2), wx};
}
DEV int lerp_pairs8i(uint32_t top, uint32_t bot, int wx, int wy) /* two 2-texel loads of a 1-channel plane: the filtered code */
{
    const int a = lerp8s((int)(top & 0xff), (int)(top >> 8), wx), b = lerp8s((int)(bot & 0xff), (int)(bot >> 8), wx);
    return lerp8s(a, b, wy);
}
DEV float lerp_pairs8(uint32_t top, uint32_t bot, int wx, int wy) { return (float)lerp_pairs8i(top, bot, wx, wy); }
/* Two lerp8s at once on words of the form 0x00XX00XX (one 8-bit sample per 16-bit half) that share a weight - Cb and Cr of
 * a pixel always do: a + (((b - a) * w + 128) >> 8) = (a * (256 - w) + b * w + 128) >> 8, the sum is at most 255 * 256 +
 * 128 < 2^16, so the halves never carry into each other and each is exactly lerp8s of its samples (weight 256 of the
 * edge trick included). UNSHIFTED: the two results sit in bytes 1 and 3. */
DEV uint32_t lerp8x2_raw(uint32_t a, uint32_t b, uint32_t w, uint32_t iw)
{
    return __umul24(a, iw) + __umul24(b, w) + 0x00800080u;
}
/* the Cb and Cr samples of a pixel from their four tap pairs (two texels per 16-bit load, zero-extended) */
DEV void lerp_chroma8(uint32_t cbt, uint32_t cbb, uint32_t crt, uint32_t crb, int wx, int wy, int& Cb, int& Cr)
{
    const uint32_t w = (uint32_t)wx, iw = 256u - w;
    const uint32_t top = (lerp8x2_raw(__builtin_amdgcn_perm(crt, cbt, 0x0c040c00u), __builtin_amdgcn_perm(crt, cbt, 0x0c050c01u), w, iw) >> 8) & 0x00FF00FFu;
    const uint32_t bot = (lerp8x2_raw(__builtin_amdgcn_perm(crb, cbb, 0x0c040c00u), __builtin_amdgcn_perm(crb, cbb, 0x0c050c01u), w, iw) >> 8) & 0x00FF00FFu;
    const uint32_t v   = lerp8x2_raw(top, bot, (uint32_t)wy, 256u - (uint32_t)wy);
    Cb = (int)((v >> 8) & 0xffu), Cr = (int)(v >> 24);
}
template <bool PACKED, bool UNIT>
DEV void fetch_rect8(const KOp& op, const HotG& g, int ax0, int ax1, int px0, int y, f4 col[4], uint32_t& cover)
{
    /* UNIT: a 1:1 source whose offset or width is off the 4-texel grid of the aligned bodies; plane 0 is addressed by
       texel index, s and t stand for texel + 0.5 and the sizes for the plane's size relative to plane 0's
       (source_color_class<true>: rint((p + .5) * ((size / size0) * 256)) - 128). Otherwise normalised coordinates from
       the quad's affine map. */
    const KTri& T  = op.tri[0];
    auto coord = [](float v, float size256) -> int { return (int)__builtin_rintf(v * size256) - 128; };
    /* t and what follows from it: computed by every lane, identical in all of them, moved to scalar registers */
    const float  t   = UNIT ? (float)(y + g.unit_dy) + 0.5f : (float)__builtin_fma(T.a[1][2], (double)y, T.a[1][0]);
    const float  sh0 = UNIT ? (g.fh0 / g.fh0) * 256.0f : g.fh0 * 256.0f, sh1 = UNIT ? (g.fh1 / g.fh0) * 256.0f : g.fh1 * 256.0f;
    const float  sw0 = UNIT ? (g.fw0 / g.fw0) * 256.0f : g.fw0 * 256.0f, sw1 = UNIT ? (g.fw1 / g.fw0) * 256.0f : g.fw1 * 256.0f;
    const int    iv0 = __builtin_amdgcn_readfirstlane(coord(t, sh0));
    const double a01 = T.a[0][1], a00 = T.a[0][0];
    const float  sc  = 1.0f / 255.0f;
    float        s[4];
    static_for<0, 4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int     x = px0 + i;
        s[i]            = UNIT ? (float)(x + g.unit_dx) + 0.5f : (float)__builtin_fma(a01, (double)x, a00);
        cover |= ((x >= ax0) & (x < ax1)) ? 1u << i : 0u;
    });
    if (PACKED) {
        const RowTaps  r   = row_taps(g.ptr0, iv0, g.w0, g.h0, 4u);
        const int      pf  = g.pixel_format;
        const uint32_t sel = pf == 1 ? 0x03020100u : pf == 2 ? 0x03000102u : pf == 3 ? 0x00010203u : 0x02030001u;
        ColTaps        c[4];
        u32x2          top[4], bot[4];
        static_for<0, 4>([&](auto I) { c[decltype(I)::value] = col_taps(coord(s[decltype(I)::value], sw0), g.w0); });
        static_for<0, 4>([&](auto I) { /* all loads first: one round trip per op */
            constexpr int i = decltype(I)::value;
            top[i] = load_at<u32x2>(r.r0, c[i].x * 4u), bot[i] = load_at<u32x2>(r.r1, c[i].x * 4u);
        });
        static_for<0, 4>([&](auto I) {
            constexpr int  i  = decltype(I)::value;
            const uint32_t wx = (uint32_t)c[i].wx;
            const uint32_t m  = lerp8x4(lerp8x4(top[i].x, top[i].y, wx), lerp8x4(bot[i].x, bot[i].y, wx), (uint32_t)r.wy);
            const uint32_t u  = __builtin_amdgcn_perm(m, m, sel);
            col[i] = f4{(float)(u & 0xff) * sc, (float)((u >> 8) & 0xff) * sc, (float)((u >> 16) & 0xff) * sc, (float)(u >> 24) * sc};
        });
    } else {
        const int     iv1 = __builtin_amdgcn_readfirstlane(coord(t, sh1));
        const RowTaps ry = row_taps(g.ptr0, iv0, g.w0, g.h0, 1u);
        const RowTaps rb = row_taps(g.ptr1, iv1, g.w1, g.h1, 1u), rr = row_taps(g.ptr2, iv1, g.w1, g.h1, 1u);
        const bool    alpha = g.pixel_format == 6;
        const bool    hd = g.flags & KF_IS_HD;
        ColTaps       c0[4], c1[4];
        uint32_t      yt[4], yb[4], bt[4], bb[4], rt[4], rbm[4];
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            c0[i] = col_taps(coord(s[i], sw0), g.w0), c1[i] = col_taps(coord(s[i], sw1), g.w1);
        });
        static_for<0, 4>([&](auto I) { /* all loads first: one round trip per op */
            constexpr int i = decltype(I)::value;
            yt[i] = load_at<uint16_t>(ry.r0, c0[i].x), yb[i] = load_at<uint16_t>(ry.r1, c0[i].x);
            bt[i] = load_at<uint16_t>(rb.r0, c1[i].x), bb[i] = load_at<uint16_t>(rb.r1, c1[i].x);
            rt[i] = load_at<uint16_t>(rr.r0, c1[i].x), rbm[i] = load_at<uint16_t>(rr.r1, c1[i].x);
        });
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int     Y = lerp_pairs8i(yt[i], yb[i], c0[i].wx, ry.wy);
            int           Cb, Cr;
            lerp_chroma8(bt[i], bb[i], rt[i], rbm[i], c1[i].wx, rb.wy, Cb, Cr);
            col[i] = ycbcr_codes_to_rgba(Y, Cb, Cr, 1.0f, hd ? 1 : 0);
        });
        if (alpha) { /* wave-uniform, and rare (ycbcra): a second round trip of its own, so that the eight registers of
                        its taps are not held (as zeros) through the three planes every ycbcr box has */
            const RowTaps ra = row_taps(g.ptr3, iv0, g.w0, g.h0, 1u);
            uint32_t      at[4], ab[4];
            static_for<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                at[i] = load_at<uint16_t>(ra.r0, c0[i].x), ab[i] = load_at<uint16_t>(ra.r1, c0[i].x);
            });
            static_for<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                col[i].w = lerp_pairs8(at[i], ab[i], c0[i].wx, ry.wy) * sc;
            });
        }
    }
}

/* FB_RECT_UYVY8: a box of an 8-bit uyvy frame (a live SDI input in a picture-in-picture box): luma pairs from the RG8
 * view, chroma pairs from the RGBA8 view of half the width; same structure and arithmetic as fetch_rect8 */
template <bool UNIT>
DEV void fetch_rect_uyvy8(const KOp& op, const HotG& g, int ax0, int ax1, int px0, int y, f4 col[4], uint32_t& cover)
{
    const KTri& T  = op.tri[0];
    auto coord = [](float v, float size256) -> int { return (int)__builtin_rintf(v * size256) - 128; };
    const float  t   = UNIT ? (float)(y + g.unit_dy) + 0.5f : (float)__builtin_fma(T.a[1][2], (double)y, T.a[1][0]);
    const float  sh0 = UNIT ? (g.fh0 / g.fh0) * 256.0f : g.fh0 * 256.0f, sh1 = UNIT ? (g.fh1 / g.fh0) * 256.0f : g.fh1 * 256.0f;
    const float  sw0 = UNIT ? (g.fw0 / g.fw0) * 256.0f : g.fw0 * 256.0f, sw1 = UNIT ? (g.fw1 / g.fw0) * 256.0f : g.fw1 * 256.0f;
    const int    iv0 = __builtin_amdgcn_readfirstlane(coord(t, sh0)), iv1 = __builtin_amdgcn_readfirstlane(coord(t, sh1));
    const double a01 = T.a[0][1], a00 = T.a[0][0];
    const RowTaps ry = row_taps(g.ptr0, iv0, g.w0, g.h0, 2u), rc = row_taps(g.ptr1, iv1, g.w1, g.h1, 4u);
    const bool    hd = g.flags & KF_IS_HD;
    ColTaps       c0[4], c1[4];
    uint32_t      yt[4], yb[4];
    u32x2         ct[4], cb[4];
    static_for<0, 4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int     x = px0 + i;
        const float   s = UNIT ? (float)(x + g.unit_dx) + 0.5f : (float)__builtin_fma(a01, (double)x, a00);
        c0[i] = col_taps(coord(s, sw0), g.w0), c1[i] = col_taps(coord(s, sw1), g.w1);
        cover |= ((x >= ax0) & (x < ax1)) ? 1u << i : 0u;
    });
    static_for<0, 4>([&](auto I) { /* all loads first */
        constexpr int i = decltype(I)::value;
        yt[i] = load_at<uint32_t>(ry.r0, c0[i].x * 2u), yb[i] = load_at<uint32_t>(ry.r1, c0[i].x * 2u);
        ct[i] = load_at<u32x2>(rc.r0, c1[i].x * 4u), cb[i] = load_at<u32x2>(rc.r1, c1[i].x * 4u);
    });
    static_for<0, 4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int     wx0 = c0[i].wx, wx1 = c1[i].wx;
        const int     Y = lerp8s(lerp8s((yt[i] >> 8) & 0xff, yt[i] >> 24, wx0), lerp8s((yb[i] >> 8) & 0xff, yb[i] >> 24, wx0), ry.wy);
        /* U and V sit in bytes 0 and 2 of a texel of the RGBA8 view: both lerped at once (lerp8x2_raw) */
        const uint32_t w1 = (uint32_t)wx1, iw1 = 256u - w1, m = 0x00FF00FFu;
        const uint32_t top = (lerp8x2_raw(ct[i].x & m, ct[i].y & m, w1, iw1) >> 8) & m;
        const uint32_t bot = (lerp8x2_raw(cb[i].x & m, cb[i].y & m, w1, iw1) >> 8) & m;
        const uint32_t uv  = lerp8x2_raw(top, bot, (uint32_t)rc.wy, 256u - (uint32_t)rc.wy);
        col[i] = ycbcr_codes_to_rgba(Y, (int)((uv >> 8) & 0xffu), (int)(uv >> 24), 1.0f, hd ? 1 : 0);
    });
}

/* FB_RECT16: the same for 16-bit planes, filtered with fp32 weights like source_color_class16<false> (make_taps_f,
 * bilerp_f), term for term. The two column taps again arrive in one load; at the plane's edges, where CLAMP_TO_EDGE
 * makes both taps the same texel, the pair's other half is overwritten with it (a + w * (b - a) with b == a is a for any
 * w; a weight of 1 would not give b exactly). */
struct RowTapsF { /* wave-uniform */
    GB    r0, r1;
    float wy;
};
DEV RowTapsF row_taps_f(uint64_t ptr, float v, int w, int h, uint32_t texel_bytes)
{
    const float fy = __builtin_floorf(v);
    int         y0 = (int)fy;
    const int   y1 = iclamp(y0 + 1, 0, h - 1);
    y0             = iclamp(y0, 0, h - 1);
    const uint32_t pitch = (uint32_t)w * texel_bytes;
    return RowTapsF{(GB)ptr + (size_t)((uint32_t)y0 * pitch), (GB)ptr + (size_t)((uint32_t)y1 * pitch), v - fy};
}
struct ColTapsF {
    uint32_t x;    /* column of the pair */
    int      edge; /* -1: both taps are the pair's first texel, +1: both are its second, 0: the pair as loaded */
    uint32_t sel;  /* the same as a byte selector for a pair of 16-bit texels in one dword */
    float    wx;
};
DEV ColTapsF col_taps_f(float u, int w)
{
    const float fx = __builtin_floorf(u);
    const int   x0 = (int)fx;
    const bool  first = x0 < 0, second = x0 > w - 2;
    return ColTapsF{(uint32_t)iclamp(x0, 0, w - 2), first ? -1 : (second ? 1 : 0),
                    first ? 0x01000100u : (second ? 0x03020302u : 0x03020100u), u - fx};
}
DEV float lerp_pairs16(uint32_t top, uint32_t bot, const ColTapsF c, float wy)
{
    const float    sc = 1.0f / 65535.0f;
    const uint32_t t = __builtin_amdgcn_perm(top, top, c.sel), b = __builtin_amdgcn_perm(bot, bot, c.sel);
    return bilerp_f((float)(t & 0xffff) * sc, (float)(t >> 16) * sc, (float)(b & 0xffff) * sc, (float)(b >> 16) * sc, c.wx, wy);
}
template <bool PACKED, bool UNIT>
DEV void fetch_rect16(const KOp& op, const HotG& g, int ax0, int ax1, int px0, int y, f4 col[4], uint32_t& cover)
{
    const KTri& T = op.tri[0];
    auto coord = [](float v, float size) -> float { return v * size - 0.5f; };
    const float  t   = UNIT ? (float)(y + g.unit_dy) + 0.5f : (float)__builtin_fma(T.a[1][2], (double)y, T.a[1][0]);
    const float  sh0 = UNIT ? g.fh0 / g.fh0 : g.fh0, sh1 = UNIT ? g.fh1 / g.fh0 : g.fh1;
    const float  sw0 = UNIT ? g.fw0 / g.fw0 : g.fw0, sw1 = UNIT ? g.fw1 / g.fw0 : g.fw1;
    const float  v0  = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, coord(t, sh0))));
    const double a01 = T.a[0][1], a00 = T.a[0][0];
    float        s[4];
    static_for<0, 4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int     x = px0 + i;
        s[i]            = UNIT ? (float)(x + g.unit_dx) + 0.5f : (float)__builtin_fma(a01, (double)x, a00);
        cover |= ((x >= ax0) & (x < ax1)) ? 1u << i : 0u;
    });
    if (PACKED) {
        const RowTapsF r = row_taps_f(g.ptr0, v0, g.w0, g.h0, 8u);
        ColTapsF       c[4];
        u32x4          top[4], bot[4];
        static_for<0, 4>([&](auto I) { c[decltype(I)::value] = col_taps_f(coord(s[decltype(I)::value], sw0), g.w0); });
        static_for<0, 4>([&](auto I) { /* all loads first */
            constexpr int i = decltype(I)::value;
            top[i] = load_at<u32x4>(r.r0, c[i].x * 8u), bot[i] = load_at<u32x4>(r.r1, c[i].x * 8u);
        });
        const float sc = 1.0f / 65535.0f;
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const bool    first = c[i].edge < 0, second = c[i].edge > 0;
            const u32x2   a00v = {second ? top[i].z : top[i].x, second ? top[i].w : top[i].y};
            const u32x2   a01v = {first ? top[i].x : top[i].z, first ? top[i].y : top[i].w};
            const u32x2   a10v = {second ? bot[i].z : bot[i].x, second ? bot[i].w : bot[i].y};
            const u32x2   a11v = {first ? bot[i].x : bot[i].z, first ? bot[i].y : bot[i].w};
#define CH(v, k) ((float)((k) == 0 ? (v).x & 0xffff : (k) == 1 ? (v).x >> 16 : (k) == 2 ? (v).y & 0xffff : (v).y >> 16) * sc)
#define LERP(k) bilerp_f(CH(a00v, k), CH(a01v, k), CH(a10v, k), CH(a11v, k), c[i].wx, r.wy)
            const float m0 = LERP(0), m1 = LERP(1), m2 = LERP(2), m3 = LERP(3);
#undef LERP
#undef CH
            col[i] = swizzle_format(g.pixel_format, f4{m2, m1, m0, m3});
        });
    } else {
        const float    v1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, coord(t, sh1))));
        const RowTapsF ry = row_taps_f(g.ptr0, v0, g.w0, g.h0, 2u);
        const RowTapsF rb = row_taps_f(g.ptr1, v1, g.w1, g.h1, 2u), rr = row_taps_f(g.ptr2, v1, g.w1, g.h1, 2u);
        const bool     alpha = g.pixel_format == 6;
        const bool     hd = g.flags & KF_IS_HD;
        ColTapsF       c0[4], c1[4];
        uint32_t       yt[4], yb[4], bt[4], bb[4], rt[4], rbm[4];
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            c0[i] = col_taps_f(coord(s[i], sw0), g.w0), c1[i] = col_taps_f(coord(s[i], sw1), g.w1);
        });
        static_for<0, 4>([&](auto I) { /* all loads first */
            constexpr int i = decltype(I)::value;
            yt[i] = load_at<uint32_t>(ry.r0, c0[i].x * 2u), yb[i] = load_at<uint32_t>(ry.r1, c0[i].x * 2u);
            bt[i] = load_at<uint32_t>(rb.r0, c1[i].x * 2u), bb[i] = load_at<uint32_t>(rb.r1, c1[i].x * 2u);
            rt[i] = load_at<uint32_t>(rr.r0, c1[i].x * 2u), rbm[i] = load_at<uint32_t>(rr.r1, c1[i].x * 2u);
        });
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const float   Y = lerp_pairs16(yt[i], yb[i], c0[i], ry.wy);
            const float   Cb = lerp_pairs16(bt[i], bb[i], c1[i], rb.wy), Cr = lerp_pairs16(rt[i], rbm[i], c1[i], rb.wy);
            col[i] = ycbcra_to_rgba(Y, Cb, Cr, 1.0f, hd);
        });
        if (alpha) { /* wave-uniform and rare: its own round trip (see fetch_rect8) */
            const RowTapsF ra = row_taps_f(g.ptr3, v0, g.w0, g.h0, 2u);
            uint32_t       at[4], ab[4];
            static_for<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                at[i] = load_at<uint32_t>(ra.r0, c0[i].x * 2u), ab[i] = load_at<uint32_t>(ra.r1, c0[i].x * 2u);
            });
            static_for<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                col[i].w = lerp_pairs16(at[i], ab[i], c0[i], ry.wy);
            });
        }
    }
}

/* FULL adds the two RGBA temporaries (layer texture for non-normal blend modes, mix texture for MIX transitions:
 * image_mixer.cpp:128,187) as 32 more registers per lane, composite draws of them (KOP_DRAW_SURFACE, any blend mode,
 * either keyer) and item draws into them. 128 VGPRs: four waves per SIMD. */
/* The aligned unit-scale fetch bodies and the adjustments as functions, for the step of a fused MIX pair
 * (composite_fast_body, MIXP), which runs them twice. The op loop itself keeps its inline switch: as a function called
 * from there the bodies were allocated differently and the headline frame paid for it. */
template <int BITS>
DEV void fetch_aligned(const KOp& op, const HotOp& h, int px0, int y, f4 col[4])
{
    const int  sx = px0 + h.unit_dx, sy = opaque_s(y + h.unit_dy);
    const bool red_only = h.flags & KF_RED_ONLY;
#define PLANAR16(HX, HY)                                                                                               \
    if constexpr (BITS == 16)                                                                                          \
        fetch_planar16r<HX, HY>(op, h, sx, sy, red_only, col);                                                         \
    else                                                                                                               \
        fetch_planar<16, HX, HY>(h, sx, sy, red_only, col);
    switch (h.fetch_body) {
        case FB_UYVY8: fetch_uyvy8(h, sx, sy, col); break;
        case FB_BYTES8: fetch_bytes8(h, sx, sy, col); break;
        case FB_PACKED8: fetch_packed<8>(h, sx, sy, col); break;
        case FB_PACKED16: fetch_packed<16>(h, sx, sy, col); break;
        case FB_PLANAR8 + 0: fetch_planar<8, 0, 0>(h, sx, sy, red_only, col); break;
        case FB_PLANAR8 + 1: fetch_planar<8, 1, 0>(h, sx, sy, red_only, col); break;
        case FB_PLANAR8 + 2: fetch_planar<8, 0, 1>(h, sx, sy, red_only, col); break;
        case FB_PLANAR8 + 3: fetch_planar<8, 1, 1>(h, sx, sy, red_only, col); break;
        case FB_PLANAR16 + 0: PLANAR16(0, 0) break;
        case FB_PLANAR16 + 1: PLANAR16(1, 0) break;
        case FB_PLANAR16 + 2: PLANAR16(0, 1) break;
        default: PLANAR16(1, 1) break;
    }
#undef PLANAR16
}
/* shader.frag:526-531 on a lane's four pixels, then the box edge inside the lane (KF_EDGE) */
DEV void adjust4(const KOp& op, const HotOp& h, int px0, f4 col[4])
{
    if (h.flags & KF_CHROMA) {
        const ChromaV cp(op);
        const bool    mask = cp.flags & KF_CHROMA_MASK, plain = !(opaque_s((int)cp.flags) & KF_CHROMA_MASK);
        if constexpr (RXF) {
            const ChromaRx k(cp);
            if (mask)
                static_for<0, 4>([&](auto I) { col[decltype(I)::value] = chroma_key_relaxed<true>(col[decltype(I)::value], k); });
            if (plain)
                static_for<0, 4>([&](auto I) { PX_BRANCH col[decltype(I)::value] = chroma_key_relaxed<false>(col[decltype(I)::value], k); });
        } else {
            if (mask)
                static_for<0, 4>([&](auto I) { col[decltype(I)::value] = chroma_key<false, 1>(col[decltype(I)::value], cp); });
            if (plain)
                static_for<0, 4>([&](auto I) { PX_BRANCH col[decltype(I)::value] = chroma_key<false, 0>(col[decltype(I)::value], cp); });
        }
    }
    if (h.flags & KF_LEVELS) {
        const LevelsV lp(op);
        levels12<RXF>(col, lp);
    }
    if (h.flags & KF_CSB) {
        const CsbV bp(op);
        static_for<0, 4>([&](auto I) { csb<RXF>(col[decltype(I)::value], bp); });
    }
    if (h.flags & KF_EDGE) {
        const uint32_t span = (uint32_t)(h.ax1 - h.ax0);
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            col[i] = sel4((uint32_t)(px0 + i - h.ax0) < span, col[i], f4{0, 0, 0, 0});
        });
    }
}

template <int BITS, bool FULL, int EXT, bool ONE = false, int MONO = -1, bool NOMODE = false, bool MIXP = false> /* EXT: 0 aligned bodies only, 1 + sampled boxes,
                                                            2 + fused blend-mode layers. ONE (with FULL): the launch never
                                                            touches RGBA2 - a MIX transition outside blend-mode layers -
                                                            and the kernel does not hold it: 16 registers less per lane.
                                                            MONO >= 0: every draw of the launch takes THIS fetch body (a
                                                            stack of sources of one format, KSpill::mono): the switch
                                                            over the bodies folds away, see cghip_composite_fast_m */
DEV void composite_fast_body(const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, const KSpill& sp)
{
    constexpr int  P       = 4;
    constexpr bool SAMPLED = EXT >= 1;
    constexpr bool FUSED   = FULL ? EXT >= 1 : EXT >= 2; /* the kernels with temporaries call the blend modes anyway */
#if CGHIP_YUV_LUT
    if constexpr (BITS == 8 && (MONO < 0 || (MONO >= FB_PLANAR8 && MONO < FB_PLANAR8 + 4) || MONO == FB_UYVY8))
        if (sp.yuv_lut) /* a kernel argument: the whole workgroup or none. Before any return: it ends with a barrier */
            fill_yuv_lut();
#endif
    const int      lane = threadIdx.x & 63;
    const int      wave = threadIdx.x >> 6;
    unsigned       bx, by;
    if (!xcd_block(sp, bx, by))
        return;
    const int y = __builtin_amdgcn_readfirstlane((int)by * (int)(blockDim.x >> 6) + wave); /* wave-uniform: keep it scalar */
    if (y >= H)
        return;
    const int px0 = (int)bx * 64 * P + lane * P;

    f4    S[P], S1[P], S2[P];
    float K0[P], K1[P];
    static_for<0, P>([&](auto I) {
        constexpr int i = decltype(I)::value;
        S[i] = S1[i] = S2[i] = f4{0, 0, 0, 0};
        K0[i] = K1[i] = 0.0f;
    });
    if (sp.load_mask != 0) { /* surfaces an earlier launch of this frame left in HBM (always the full grid here) */
        static_for<0, P>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (px0 + i < W) {
                const size_t idx = (size_t)y * W + px0 + i;
                if (sp.load_mask & (1u << KS_RGBA0)) S[i] = load_px<BITS>(out, idx);
                if (FULL && (sp.load_mask & (1u << KS_RGBA1))) S1[i] = load_px<BITS>(sp.rgba1, idx);
                if (FULL && !ONE && (sp.load_mask & (1u << KS_RGBA2))) S2[i] = load_px<BITS>(sp.rgba2, idx);
                if (sp.load_mask & (1u << KS_KEY0)) K0[i] = load_key<BITS>(sp.key0, idx);
                if (sp.load_mask & (1u << KS_KEY1)) K1[i] = load_key<BITS>(sp.key1, idx);
            }
        });
    }
    const bool row_live = !((field_mode == 1 && (y & 1) != 0) || (field_mode == 2 && (y & 1) != 1));

    /* only the ops that can touch this wave's band of the frame (KSpill::masks, host: op_masks) */
    const int mcol = __builtin_amdgcn_readfirstlane(px0) >> sp.col_shift, mrow = y >> 2;
    const int mstride = sp.ncols + sp.nrows;
    int       mix_skip = 0; /* MIXP: ops still to pass over behind a BB_MIX_HEAD */
    (void)mix_skip;
    for (int chunk = 0; chunk < sp.nchunks && row_live; ++chunk)
    for (uint64_t todo = band_ops(sp, chunk, mstride, mcol, mrow, nops); todo != 0; todo &= todo - 1) {
        const int   n   = chunk * 64 + __builtin_ctzll(todo);
        const KOp&  op  = ops[__builtin_amdgcn_readfirstlane(n)]; /* n is uniform; say so (see dispatch_mode) */
        const HotOp h   = load_hot(op);
        if constexpr (MIXP) { /* the two ops behind a BB_MIX_HEAD were run by its step (wave-uniform) */
            if (mix_skip > 0) {
                --mix_skip;
                continue;
            }
            if (h.kind == KOP_DRAW_PLANES && h.blend_body == BB_MIX_HEAD)
                mix_skip = 2;
        }
        const bool  clr = h.kind == KOP_CLEAR;
        const bool  clr0 = clr & (h.dst == KS_KEY0), clr1 = clr & (h.dst == KS_KEY1);
        if (clr) { /* wave-uniform; the selects stay inside so that draws do not pay for them */
            static_for<0, P>([&](auto I) {
                constexpr int i = decltype(I)::value;
                K0[i] = clr0 ? 0.0f : K0[i];
                K1[i] = clr1 ? 0.0f : K1[i];
                if (FULL) {
                    S1[i] = sel4(h.dst == KS_RGBA1, f4{0, 0, 0, 0}, S1[i]);
                    if constexpr (!ONE)
                        S2[i] = sel4(h.dst == KS_RGBA2, f4{0, 0, 0, 0}, S2[i]);
                }
            });
        }
        /* one combined test, one merge point: a chain of early `continue`s made the compiler copy all 24
           loop-carried registers at every exit. A lane takes part if any of its 4 pixels is in the box; x0, x1 are
           multiples of 4 (a lane is all in or all out) unless the op carries KF_EDGE, see below. */
        if (clr | !((y >= h.y0) & (y < h.y1) & (px0 > h.x0 - P) & (px0 < h.x1)))
            continue;
        if constexpr (MIXP) {
            if (h.blend_body == BB_MIX_HEAD) {
                /* A fused MIX pair (cghip_plan.cpp: fuse_mix_pairs): this op and the next are the two items drawn into
                   the cleared mix texture, the one after composites it (image_mixer.cpp:187-225). The texture T lives in
                   this step only. Each item: fore colour, then T = store(fore + ia * T) - store_rgba4 over the zero T
                   for the first, exactly what the kernels with a temporary do op by op. */
                f4 T[P];
                static_for<0, P>([&](auto I) { T[decltype(I)::value] = f4{0, 0, 0, 0}; });
                auto item = [&](const KOp& o, const HotOp& g) {
                    f4 c[P];
                    fetch_aligned<BITS>(o, g, px0, y, c);
                    adjust4(o, g, px0, c);
                    Fore fo;
                    fore4<2>(g, c, K0, K1, fo);
                    store_rgba4<BITS>(fo, T);
                };
                item(op, h);
                const KOp& opb = ops[__builtin_amdgcn_readfirstlane(n + 1)];
                item(opb, load_hot(opb));
                const KOp&  ops_ = ops[__builtin_amdgcn_readfirstlane(n + 2)];
                const HotOp hs   = load_hot(ops_);
                Fore        fo;
                switch (hs.blend_body) {
                    case BB_RGBA_SIMPLE: fore4<0>(hs, T, K0, K1, fo); break;
                    case BB_RGBA_ONE_KEY: fore4<1>(hs, T, K0, K1, fo); break;
                    default: fore4<2>(hs, T, K0, K1, fo); break;
                }
                if constexpr (FULL) { /* inside a blend-mode layer the composite's target is the layer texture (two ifs: see below) */
                    if (hs.dst != KS_RGBA0)
                        store_rgba4<BITS>(fo, S1);
                    if (opaque_s(hs.dst) == KS_RGBA0)
                        store_rgba4<BITS>(fo, S);
                } else if constexpr (FUSED) {
                    /* the pair is all its blend-mode layer holds (KOp::fused on the composite): over the cleared layer
                       texture the composite leaves its fore colour through the UNORM store, and that texel goes through
                       the layer's blend mode onto the frame, like a fused single item (BB_FUSED below) */
                    const int lf = ops_.fused;
                    if (lf) {
                        f4 tex[P];
                        static_for<0, P>([&](auto I) {
                            constexpr int i = decltype(I)::value;
                            tex[i] = f4{quantize<BITS>(fo.f[i].x), quantize<BITS>(fo.f[i].y), quantize<BITS>(fo.f[i].z), quantize<BITS>(fo.f[i].w)};
                        });
                        shade_blend_mode4<BITS>(BlendArgs{ops_.b_flags, ops_.b_local_key, ops_.b_layer_key, ops_.b_opacity, ops_.b_blend_mode}, tex, S, K0, K1);
                    }
                    if (!opaque_s(lf))
                        store_rgba4<BITS>(fo, S);
                } else {
                    store_rgba4<BITS>(fo, S);
                }
                continue;
            }
        }
        /* Source colour: the body the planner picked (KOp::fetch_body). Each body starts from an OPAQUE copy of the
           source row: left to itself the compiler hoists the row-address arithmetic of all ten bodies in front of the
           switch, where every op and wave paid ~90 scalar instructions for bodies it does not run - and the scalar
           unit is shared by the CU's four SIMDs. */
        const int  sx = px0 + h.unit_dx;
        const bool red_only = h.flags & KF_RED_ONLY;
        f4         col[P];
        uint32_t   cover = 0; /* sampled bodies: which of the lane's pixels the quad covers */
        if constexpr (!SAMPLED) /* the sampled bodies are empty in this instantiation (never planned for it) */
            static_for<0, P>([&](auto I) { col[decltype(I)::value] = f4{0, 0, 0, 0}; });
#define SY opaque_s(y + h.unit_dy)
#define PLANAR16(HX, HY)                                                                                               \
    if constexpr (BITS == 16)                                                                                          \
        fetch_planar16r<HX, HY>(op, h, sx, SY, red_only, col);                                                         \
    else                                                                                                               \
        fetch_planar<16, HX, HY>(h, sx, SY, red_only, col);
        switch (MONO >= 0 ? ((FULL && h.fetch_body == FB_SURFACE) ? (int)FB_SURFACE : MONO) : h.fetch_body) {
            case FB_SURFACE:
                if constexpr (FULL) {
                    if constexpr (ONE) {
                        static_for<0, P>([&](auto I) { col[decltype(I)::value] = S1[decltype(I)::value]; });
                    } else {
                        const bool from2 = op.src_surface == KS_RGBA2;
                        static_for<0, P>([&](auto I) { col[decltype(I)::value] = sel4(from2, S2[decltype(I)::value], S1[decltype(I)::value]); });
                    }
                } else {
                    static_for<0, P>([&](auto I) { col[decltype(I)::value] = f4{0, 0, 0, 0}; }); /* never planned */
                }
                break;
            /* sampled bodies: like the aligned ones, each starts from opaque copies of its row and column, or the
               compiler evaluates every body's coordinate arithmetic in front of the switch for every op */
            case FB_RECT8 + 0:
                if constexpr (SAMPLED)
                    fetch_rect8<false, false>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_RECT8 + 1:
                if constexpr (SAMPLED)
                    fetch_rect8<true, false>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_RECT8 + 2:
                if constexpr (SAMPLED)
                    fetch_rect8<false, true>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_RECT8 + 3:
                if constexpr (SAMPLED)
                    fetch_rect8<true, true>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_RECT_UYVY8 + 0:
                if constexpr (SAMPLED)
                    fetch_rect_uyvy8<false>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_RECT_UYVY8 + 1:
                if constexpr (SAMPLED)
                    fetch_rect_uyvy8<true>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_RECT16 + 0:
                if constexpr (SAMPLED)
                    fetch_rect16<false, false>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_RECT16 + 1:
                if constexpr (SAMPLED)
                    fetch_rect16<true, false>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_RECT16 + 2:
                if constexpr (SAMPLED)
                    fetch_rect16<false, true>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_RECT16 + 3:
                if constexpr (SAMPLED)
                    fetch_rect16<true, true>(op, load_hot_g<false>(op), h.ax0, h.ax1, opaque_v(px0), opaque_s(y), col, cover);
                break;
            case FB_UYVY8: fetch_uyvy8(h, sx, SY, col); break;
            case FB_BYTES8: fetch_bytes8(h, sx, SY, col); break;
            case FB_PACKED8: fetch_packed<8>(h, sx, SY, col); break;
            case FB_PACKED16: fetch_packed<16>(h, sx, SY, col); break;
            case FB_PLANAR8 + 0: fetch_planar<8, 0, 0>(h, sx, SY, red_only, col); break;
            case FB_PLANAR8 + 1: fetch_planar<8, 1, 0>(h, sx, SY, red_only, col); break;
            case FB_PLANAR8 + 2: fetch_planar<8, 0, 1>(h, sx, SY, red_only, col); break;
            case FB_PLANAR8 + 3: fetch_planar<8, 1, 1>(h, sx, SY, red_only, col); break;
            case FB_PLANAR16 + 0: PLANAR16(0, 0) break;
            case FB_PLANAR16 + 1: PLANAR16(1, 0) break;
            case FB_PLANAR16 + 2: PLANAR16(0, 1) break;
            default: PLANAR16(1, 1) break;
        }
#undef PLANAR16
#undef SY
        /* shader.frag:526-531, each adjustment one wave-uniform branch around all four pixels */
        if (h.flags & KF_CHROMA) {
            const ChromaV cp(op);
            /* two ifs with the second test opaque, not if / else: see the stores below */
            const bool mask = cp.flags & KF_CHROMA_MASK, plain = !(opaque_s((int)cp.flags) & KF_CHROMA_MASK);
            if constexpr (RXF) {
                const ChromaRx k(cp);
                if (mask)
                    static_for<0, P>([&](auto I) { col[decltype(I)::value] = chroma_key_relaxed<true>(col[decltype(I)::value], k); });
                if (plain)
                    static_for<0, P>([&](auto I) { PX_BRANCH col[decltype(I)::value] = chroma_key_relaxed<false>(col[decltype(I)::value], k); });
            } else {
                if (mask)
                    static_for<0, P>([&](auto I) { col[decltype(I)::value] = chroma_key<false, 1>(col[decltype(I)::value], cp); });
                if (plain)
                    static_for<0, P>([&](auto I) { PX_BRANCH col[decltype(I)::value] = chroma_key<false, 0>(col[decltype(I)::value], cp); });
            }
        }
        if (h.flags & KF_LEVELS) {
            const LevelsV lp(op);
            levels12<RXF>(col, lp);
        }
        if (h.flags & KF_CSB) {
            const CsbV bp(op);
            static_for<0, P>([&](auto I) { csb<RXF>(col[decltype(I)::value], bp); });
        }
        if (h.flags & KF_EDGE) { /* a box edge inside a lane's 4 pixels: a zero source leaves the target as it is */
            const uint32_t span = (uint32_t)(h.ax1 - h.ax0); /* one unsigned compare per pixel: ax0 <= x < ax1 */
            static_for<0, P>([&](auto I) {
                constexpr int i = decltype(I)::value;
                col[i] = sel4((uint32_t)(px0 + i - h.ax0) < span, col[i], f4{0, 0, 0, 0});
            });
        }
        if (SAMPLED && h.fetch_body >= FB_RECT8 && h.fetch_body < FB_SAMPLED_END) { /* after the adjustments, which turn zero into something else */
            static_for<0, P>([&](auto I) {
                constexpr int i = decltype(I)::value;
                col[i] = sel4((cover >> i) & 1u, col[i], f4{0, 0, 0, 0});
            });
        }
        if constexpr (!FULL) {
            /* fore colour and background factor: the body the planner picked (KOp::blend_body) ... */
            Fore fo;
            switch (h.blend_body) {
                case BB_RGBA_SIMPLE: fore4<0>(h, col, K0, K1, fo); break;
                case BB_RGBA_ONE_KEY: fore4<1>(h, col, K0, K1, fo); break;
                case BB_KEY_SIMPLE: fore_key4<true>(h, col, K0, fo); break;
                case BB_KEY_GENERAL: fore_key4<false>(h, col, K0, fo); break;
                default: fore4<2>(h, col, K0, K1, fo); break; /* BB_FUSED too: the item's fore colour */
            }
            /* ... and ONE place per surface where its registers are rewritten */
            if (FUSED && h.blend_body == BB_FUSED) {
                /* a blend-mode layer with one item (KOp::fused): over the cleared layer texture the item's result is
                   its fore colour (fore + ia * 0), stored as UNORM; that texel is the composite's source */
                f4 tex[P];
                static_for<0, P>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    tex[i] = f4{quantize<BITS>(fo.f[i].x), quantize<BITS>(fo.f[i].y), quantize<BITS>(fo.f[i].z), quantize<BITS>(fo.f[i].w)};
                });
                shade_blend_mode4<BITS>(BlendArgs{op.b_flags, op.b_local_key, op.b_layer_key, op.b_opacity, op.b_blend_mode}, tex, S, K0, K1);
            } else {
                /* Two ifs, not if / else, and the second test opaque: the compiler structurizes uniform branches too, and
                   an if / else gets a never-taken edge from its `then` to its `else` block. Across that edge the frame
                   registers a key draw leaves alone stay live through the colour draw's block, so the colour draw could
                   not update them where they live: sixteen results in temporaries, sixteen copies at the loop latch per
                   op and lane (8 % of a plain layer's instructions). Two triangles have no such edge. */
                if (h.dst >= KS_KEY0)
                    store_key4<BITS>(fo, h.dst == KS_KEY1, K0, K1);
                if (opaque_s(h.dst) < KS_KEY0)
                    store_rgba4<BITS>(fo, S);
            }
        } else {
            /* With the two temporaries the kernel sits at its 128-register bound: carrying the fore colour of four
               pixels across a merge spilled (config 4: 0.025 -> 0.042 ms), so here every variant rewrites its surface
               itself. */
            auto blend_into = [&](f4(&D)[P]) {
                Fore fo;
                switch (h.blend_body) {
                    case BB_RGBA_SIMPLE: fore4<0>(h, col, K0, K1, fo), store_rgba4<BITS>(fo, D); break;
                    case BB_RGBA_ONE_KEY: fore4<1>(h, col, K0, K1, fo), store_rgba4<BITS>(fo, D); break;
                    case BB_MODE: /* a layer temporary composited with its blend mode (NOMODE: never planned for the instance) */
                        if constexpr (!NOMODE)
                            shade_blend_mode4<BITS>(BlendArgs{h.flags, h.local_key, h.layer_key, h.opacity, op.blend_mode}, col, D, K0, K1);
                        break;
                    case BB_FUSED: /* see the kernel without temporaries */
                        if constexpr (FUSED) {
                            fore4<2>(h, col, K0, K1, fo);
                            f4 tex[P];
                            static_for<0, P>([&](auto I) {
                                constexpr int i = decltype(I)::value;
                                tex[i] = f4{quantize<BITS>(fo.f[i].x), quantize<BITS>(fo.f[i].y), quantize<BITS>(fo.f[i].z), quantize<BITS>(fo.f[i].w)};
                            });
                            shade_blend_mode4<BITS>(BlendArgs{op.b_flags, op.b_local_key, op.b_layer_key, op.b_opacity, op.b_blend_mode}, tex, D, K0, K1);
                        }
                        break;
                    default: fore4<2>(h, col, K0, K1, fo), store_rgba4<BITS>(fo, D); break;
                }
            };
            if (h.blend_body == BB_KEY_SIMPLE) {
                Fore fo;
                fore_key4<true>(h, col, K0, fo), store_key4<BITS>(fo, h.dst == KS_KEY1, K0, K1);
            } else if (h.blend_body == BB_KEY_GENERAL) {
                Fore fo;
                fore_key4<false>(h, col, K0, fo), store_key4<BITS>(fo, h.dst == KS_KEY1, K0, K1);
            } else if (h.dst == KS_RGBA0)
                blend_into(S);
            else if (ONE || h.dst == KS_RGBA1)
                blend_into(S1);
            else
                blend_into(S2);
        }
    }

    if (px0 >= W)
        return;
    if (sp.store_mask & ~(1u << KS_RGBA0)) { /* surfaces a later launch of this frame needs */
        static_for<0, P>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (px0 + i < W) {
                const size_t idx = (size_t)y * W + px0 + i;
                if (FULL && (sp.store_mask & (1u << KS_RGBA1))) store_px<BITS>(sp.rgba1, idx, S1[i]);
                if (FULL && !ONE && (sp.store_mask & (1u << KS_RGBA2))) store_px<BITS>(sp.rgba2, idx, S2[i]);
                if (sp.store_mask & (1u << KS_KEY0)) store_key<BITS>(sp.key0_st, idx, K0[i]);
                if (sp.store_mask & (1u << KS_KEY1)) store_key<BITS>(sp.key1_st, idx, K1[i]);
            }
        });
    }
    if (!(sp.store_mask & (1u << KS_RGBA0)))
        return;
    if (BITS == 8) {
        uint32_t v[P];
        static_for<0, P>([&](auto I) {
            constexpr int i = decltype(I)::value;
            v[i] = to_code<8>(S[i].x) | (to_code<8>(S[i].y) << 8) | (to_code<8>(S[i].z) << 16) | (to_code<8>(S[i].w) << 24);
        });
        uint32_t* o = (uint32_t*)out + (size_t)y * W + px0;
        if (px0 + P <= W && (W & 3) == 0) {
            /* non-temporal unless the frame buffer is host memory (KSpill::host_out): config 1 -3 %, 16 BGRA16 layers at
               2160p -4 %, config 3 -1 %; into pinned memory the plain stores are the faster ones (16-bit: 1.28 against
               2.03 ms per frame: a lane's two 16-byte stores are merged into lines by the L2 before they cross PCIe) */
            if (sp.host_out)
                *(uint4*)o = uint4{v[0], v[1], v[2], v[3]};
            else
                __builtin_nontemporal_store(u32x4{v[0], v[1], v[2], v[3]}, (u32x4*)o);
        } else {
            static_for<0, P>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if (px0 + i < W)
                    o[i] = v[i];
            });
        }
    } else {
        uint2 v[P];
        static_for<0, P>([&](auto I) {
            constexpr int i = decltype(I)::value;
            v[i] = uint2{to_code<16>(S[i].x) | (to_code<16>(S[i].y) << 16), to_code<16>(S[i].z) | (to_code<16>(S[i].w) << 16)};
        });
        uint2* o = (uint2*)out + (size_t)y * W + px0;
        if (px0 + P <= W && (W & 1) == 0) {
            if (sp.host_out) {
                ((uint4*)o)[0] = uint4{v[0].x, v[0].y, v[1].x, v[1].y};
                ((uint4*)o)[1] = uint4{v[2].x, v[2].y, v[3].x, v[3].y};
            } else {
                __builtin_nontemporal_store(u32x4{v[0].x, v[0].y, v[1].x, v[1].y}, (u32x4*)o);
                __builtin_nontemporal_store(u32x4{v[2].x, v[2].y, v[3].x, v[3].y}, (u32x4*)o + 1);
            }
        } else {
            static_for<0, P>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if (px0 + i < W)
                    o[i] = v[i];
            });
        }
    }
}

/* The two variants as kernels of their own, each with its register bound (amdgpu_waves_per_eu(n): at least n waves
 * per SIMD, i.e. at most 512 / n VGPRs). With the temporaries: left alone 149 VGPRs (three waves), bounded 128 without a
 * spill (16 blend layers at 2160p 0.78 -> 0.65 ms). Without them: left alone 79 (six waves), bounded to seven waves 72
 * with four registers spilled outside the hot bodies (headline frame 0.285 -> 0.274 ms; eight waves, 64 VGPRs with five spilled: 0.278). */
#ifndef LEAN_WAVES
#define LEAN_WAVES 7
#endif
template <int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LEAN_WAVES))) void cghip_composite_fast(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, false, 0>(ops, nops, out, W, H, field_mode, sp);
}
template <int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void cghip_composite_fast_full(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, true, 0>(ops, nops, out, W, H, field_mode, sp);
}
/* One temporary only (KSpill::one_temp: no op of the launch touches RGBA2): MIX transitions between unit-scale sources
 * outside blend-mode layers, BASELINE config 4. */
#ifndef ONE_TEMP_WAVES
#define ONE_TEMP_WAVES 4
#endif
template <int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ONE_TEMP_WAVES))) void cghip_composite_fast_one(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, true, 0, true>(ops, nops, out, W, H, field_mode, sp);
}
#ifndef ONE_MONO_WAVES
#define ONE_MONO_WAVES 5
#endif
/* ... and its one-format instance (see cghip_composite_fast_m below): MIX transitions between bgra sources */
template <int BITS, int MONO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ONE_MONO_WAVES))) void cghip_composite_fast_one_m(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, true, 0, true, MONO, true>(ops, nops, out, W, H, field_mode, sp);
}
/* The lean kernel with the step for fused MIX pairs (BB_MIX_HEAD, KSpill::mix_pairs): crossfades between unit-scale
 * sources without a temporary. A kernel of its own: the step's registers (the mix texture of four pixels) would cost
 * every other launch of the lean kernel its seventh wave. */
#ifndef MIX_WAVES
#define MIX_WAVES 5
#endif
template <int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MIX_WAVES))) void cghip_composite_fast_p(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, false, 0, false, -1, false, true>(ops, nops, out, W, H, field_mode, sp);
}
/* Stacks of ONE source format - what a channel fed by one kind of input composites: 4:2:2 video layers with their
 * keys, an SDI rundown of uyvy inputs, stills and graphics in bgra. Every draw of such a launch takes the same fetch
 * body (KSpill::mono, cghip_plan.cpp), and an instance of the lean kernel compiled for that body alone has no switch
 * over twenty bodies in its op loop: fewer paths meet at the loop latch (fewer register copies there), less scalar
 * set-up per op. The instances below are the formats the BASELINE configs and live inputs use; anything else runs the
 * kernel with all bodies, as before. Same arithmetic, same code: the body is a template argument. */
#ifndef MONO_WAVES
/* The same step in the kernel with ONE temporary: a MIX pair inside a blend-mode layer, whose composite goes into the
 * layer texture (cghip_plan.cpp: fuse_mix_pairs leaves at most one temporary and calls it RGBA1). */
/* ... and in the kernel without temporaries that knows the blend modes: a MIX pair that is all its blend-mode layer
 * holds, handed to the layer's blend in registers (KOp::fused on the pair's composite). */
#ifndef XMIX_WAVES
#define XMIX_WAVES 4 /* mix_blend probe, 16 layers at 2160p: 0.80 / 0.77 ms (8- / 16-bit) against 0.97 / 0.81 at 5 waves, where 17 - 26 registers spill */
#endif
template <int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XMIX_WAVES))) void cghip_composite_fast_xp(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, false, 2, false, -1, false, true>(ops, nops, out, W, H, field_mode, sp);
}

#ifndef ONE_MIX_WAVES
#define ONE_MIX_WAVES 4
#endif
template <int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ONE_MIX_WAVES))) void cghip_composite_fast_one_p(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, true, 0, true, -1, false, true>(ops, nops, out, W, H, field_mode, sp);
}

#define MONO_WAVES LEAN_WAVES
#endif
template <int BITS, int MONO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MONO_WAVES))) void cghip_composite_fast_m(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, false, 0, false, MONO>(ops, nops, out, W, H, field_mode, sp);
}
/* The same two with the sampled fetch bodies (FB_RECT8) compiled in, for launches that hold such a draw.
 * Kernels of their own: with those bodies in it the kernel of full-screen stacks allocated its registers differently
 * and the headline frame went from 0.207 to 0.284 ms. */
#ifndef SAMPLED_WAVES
#define SAMPLED_WAVES 5
#endif
template <int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SAMPLED_WAVES))) void cghip_composite_fast_s(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, false, 1>(ops, nops, out, W, H, field_mode, sp);
}
/* ... and with the fused blend-mode layers (BB_FUSED, a call into the blend modes) on top: with that call compiled in,
 * plain sampled boxes ran 10 - 15 % slower (16 scaled 4:2:2 layers 0.468 -> 0.545 ms) */
template <int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SAMPLED_WAVES))) void cghip_composite_fast_x(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, false, 2>(ops, nops, out, W, H, field_mode, sp);
}
template <int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void cghip_composite_fast_full_s(
    const KOp* __restrict__ ops, int nops, void* out, int W, int H, int field_mode, KSpill sp)
{
    composite_fast_body<BITS, true, 1>(ops, nops, out, W, H, field_mode, sp);
}

#if CGHIP_FRAME_BITS == 8
/* ------------------------------------------------------------------ scratch of the allocation-free entry point */

/* cghip_mixer_render_noalloc (include/cghip.h) may not allocate, so the device memory a frame needs BEYOND its source
 * planes and its output - the op list the kernels walk, and the surfaces handed from one launch of a cut frame to the
 * next (KSpill) - lives in this file-scope array: CGHIP_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends.
 * In host code the array's name is a host-side stand-in, not a device address: its device address is looked up once
 * per device with hipGetSymbolAddress (with that device current; every GPU holds its own copy at its own address),
 * and from then on only that pointer is used - as kernel argument and as copy destination. */
static __device__ __align__(256) unsigned char scratch[CGHIP_NOALLOC_SCRATCH_BYTES];

constexpr int   SCRATCH_MAX_GPUS = 64;
static void*    g_scratch_ptr[SCRATCH_MAX_GPUS];
static std::mutex g_scratch_mu;

/* The device address of this GPU's scratch array; the first call on a GPU looks it up. No allocation. */
extern "C" hipError_t cghip_scratch_address(int gpu, void** ptr, size_t* bytes)
{
    if (gpu < 0 || gpu >= SCRATCH_MAX_GPUS || !ptr)
        return hipErrorInvalidValue;
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[gpu]) {
        hipError_t e = hipSetDevice(gpu);
        if (e != hipSuccess)
            return e;
        void* p = nullptr;
        e       = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess)
            return e;
        if (!p)
            return hipErrorInvalidSymbol;
        g_scratch_ptr[gpu] = p;
    }
    *ptr = g_scratch_ptr[gpu];
    if (bytes)
        *bytes = sizeof scratch;
    return hipSuccess;
}

#endif /* CGHIP_FRAME_BITS == 8 */

/* ------------------------------------------------------------------ launchers (called from cghip_runtime.cpp) */

constexpr unsigned XCD_MIN_TILES = 4096; /* 1080p is 2160 tiles of the 4-pixels-per-lane kernel, 2160p 8100 */
/* measurement switches, shared by both objects */
#if CGHIP_FRAME_BITS == 8
int      cghip_g_plain_tiles = 0;
int      cghip_g_pad_lds     = 0;
unsigned cghip_g_fast_rows   = 4;
int      cghip_g_no_mono     = 0;
extern "C" void cghip_debug_plain_tiles(int on) { cghip_g_plain_tiles = on; } /* tiles in dispatch order */
extern "C" void cghip_debug_no_mono_kernels(int on) { cghip_g_no_mono = on; } /* one-format stacks run the kernel with all bodies */
extern "C" void cghip_debug_fast_rows(int rows) { cghip_g_fast_rows = rows >= 1 && rows <= 4 ? (unsigned)rows : 4u; }
/* occupancy experiments: dynamic LDS nobody touches, to cap the workgroups resident on a CU (160 KB per CU) */
extern "C" void cghip_debug_pad_lds(int bytes) { cghip_g_pad_lds = bytes; }
#else
extern int      cghip_g_plain_tiles;
extern int      cghip_g_no_mono;
extern int      cghip_g_pad_lds;
extern unsigned cghip_g_fast_rows;
#endif

/* One launch of a frame of CGHIP_FRAME_BITS bits (cghip_plan.cpp: segment). sp->gx0, gy0 and gw, gh: the pixel box
 * the grid covers. */
#if CGHIP_FRAME_BITS == 8
#define LAUNCH_NAME cghip_launch_composite8
#else
#define LAUNCH_NAME cghip_launch_composite16
#endif
extern "C" hipError_t LAUNCH_NAME(const KOp* d_ops, int nops, void* d_out, int W, int H, int field_mode, int full,
                                  int fast, const KSpill* sp, int gw, int gh, hipStream_t stream)
{
    constexpr int FB = CGHIP_FRAME_BITS;
    if (gw <= 0 || gh <= 0 || sp->gx0 < 0 || sp->gy0 < 0 || sp->gx0 + gw > W || sp->gy0 + gh > H)
        return hipErrorInvalidValue;
    /* the op masks must cover the frame in the band geometry this kernel indexes them with, and every op */
    if (!sp->masks || sp->col_shift != (fast ? 8 : 6) || sp->ncols != ((W - 1) >> sp->col_shift) + 1 ||
        sp->nrows != ((H - 1) >> 2) + 1 || sp->nchunks < 1 || sp->nchunks * 64 < nops)
        return hipErrorInvalidValue;
    KSpill s = *sp;
    dim3   block(256);
    auto   tiles = [&s](unsigned nbx, unsigned nby, bool by_xcd) { /* see xcd_block */
        s.nbx = s.nby = 0;
        if (!by_xcd)
            return dim3(nbx, nby);
        s.nbx = (int32_t)nbx, s.nby = (int32_t)nby;
        const unsigned C = nbx * XCD_ROWS, chunks = (nbx * nby + C - 1) / C; /* whole chunks, as many on every XCD */
        return dim3((chunks + 7u) / 8u * 8u * C);
    };
    if (fast) {
        if (s.gx0 != 0 || s.gy0 != 0)
            return hipErrorInvalidValue; /* the 4-pixels-per-lane kernel always covers the frame */
        const unsigned rows = cghip_g_fast_rows; /* rows (= waves) per workgroup */
        block = dim3(64 * rows);
        const unsigned nbx = (unsigned)((gw + 255) / 256), nby = (unsigned)((gh + rows - 1) / rows);
        const dim3     grid = tiles(nbx, nby, nbx * nby >= XCD_MIN_TILES && !cghip_g_plain_tiles);
        const bool sampled = fast >= 2; /* the launch holds a draw with a sampled fetch body (2) or a fused layer (3) */
#define LAUNCH_MONO(body)                                                                                              \
    hipLaunchKernelGGL((cghip_composite_fast_m<FB, body>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s)
        if (s.mix_pairs) { /* planned only for launches of aligned draws with at most one temporary */
            if (full && (!s.one_temp || sampled))
                return hipErrorInvalidValue;
            if (full)
                hipLaunchKernelGGL((cghip_composite_fast_one_p<FB>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
            else if (sampled) /* a pair fused with its blend-mode layer, or sampled boxes / fused layers elsewhere in the frame */
                hipLaunchKernelGGL((cghip_composite_fast_xp<FB>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
            else
                hipLaunchKernelGGL((cghip_composite_fast_p<FB>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
            return hipGetLastError();
        }
        const int mono = (!full && !sampled && !cghip_g_no_mono) ? s.mono : -1;
        bool      launched = true; /* the instances of this object's frame depth only; measured on the same box against
                                      the kernel with all bodies (tests/gpu_ab.sh): packed 8-bit stacks -7 % (config 1
                                      -5 %), 4:2:2 16-bit stacks -1 .. -3 %. 8-bit planar and 16-bit packed instances
                                      measured 0 .. +3 % (their registers were allocated no better) and do not exist. */
        if constexpr (FB == 16) {
            if (mono == FB_PLANAR16 + 1)
                LAUNCH_MONO(FB_PLANAR16 + 1);
            else
                launched = false;
        } else {
            if (mono == FB_PACKED8)
                LAUNCH_MONO(FB_PACKED8);
            else
                launched = false;
        }
        if (launched)
            return hipGetLastError();
        if (!full && fast == 3)
            hipLaunchKernelGGL((cghip_composite_fast_x<FB>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
        else if (!full && !sampled)
            hipLaunchKernelGGL((cghip_composite_fast<FB>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
        else if (!sampled && s.one_temp == 2 && FB == 8 && s.mono == FB_PACKED8 && !cghip_g_no_mono)
            hipLaunchKernelGGL((cghip_composite_fast_one_m<FB, FB_PACKED8>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
        else if (!sampled && s.one_temp)
            hipLaunchKernelGGL((cghip_composite_fast_one<FB>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
        else if (!sampled)
            hipLaunchKernelGGL((cghip_composite_fast_full<FB>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
        else if (!full)
            hipLaunchKernelGGL((cghip_composite_fast_s<FB>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
        else
            hipLaunchKernelGGL((cghip_composite_fast_full_s<FB>), grid, block, (size_t)cghip_g_pad_lds, stream, d_ops, nops, d_out, W, H, field_mode, s);
        return hipGetLastError();
    }
    const dim3 grid = tiles((unsigned)((gw + 63) / 64), (unsigned)((gh + 3) / 4), false);
    if (full && s.fused)
        hipLaunchKernelGGL((cghip_composite<FB, true, true>), grid, block, 0, stream, d_ops, nops, d_out, W, H, field_mode, s);
    else if (full)
        hipLaunchKernelGGL((cghip_composite<FB, true>), grid, block, 0, stream, d_ops, nops, d_out, W, H, field_mode, s);
    else if (s.fused)
        hipLaunchKernelGGL((cghip_composite<FB, false, true>), grid, block, 0, stream, d_ops, nops, d_out, W, H, field_mode, s);
    else
        hipLaunchKernelGGL((cghip_composite<FB, false>), grid, block, 0, stream, d_ops, nops, d_out, W, H, field_mode, s);
    return hipGetLastError();
}

#if CGHIP_FRAME_BITS == 8
extern "C" hipError_t cghip_launch_composite16(const KOp* d_ops, int nops, void* d_out, int W, int H, int field_mode,
                                               int full, int fast, const KSpill* sp, int gw, int gh, hipStream_t stream);
extern "C" hipError_t cghip_launch_composite(const KOp* d_ops, int nops, void* d_out, int W, int H, int bits,
                                             int field_mode, int full, int fast, const KSpill* sp, int gw, int gh,
                                             hipStream_t stream)
{
    if (bits == 8)
        return cghip_launch_composite8(d_ops, nops, d_out, W, H, field_mode, full, fast, sp, gw, gh, stream);
    if (bits == 16)
        return cghip_launch_composite16(d_ops, nops, d_out, W, H, field_mode, full, fast, sp, gw, gh, stream);
    return hipErrorInvalidValue;
}
#endif
