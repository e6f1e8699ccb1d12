/*
 * oracle/ref_cpu.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Scalar CPU restatement of what the GL pipeline does for ONE reference draw:
 *   accelerator/ogl/image/shader.vert:8-16        vertex stage
 *   accelerator/ogl/image/shader.frag:48-542      fragment stage (every function restated below, cited)
 *   accelerator/ogl/util/texture.cpp:31-59        texel formats, LINEAR filter, CLAMP_TO_EDGE
 *   accelerator/ogl/image/image_kernel.cpp:309-329,378-403   viewport, scissor, two triangles
 * plus the GL semantics that are not in the reference repo (third-party: the GL implementation; the
 * goldens come from Mesa 23.2.1 llvmpipe, LLVM 15): fixed-point rasterisation with the top-left fill
 * rule, pixel-centre sampling, per-triangle affine varying interpolation (w = 1), bilinear filtering
 * with llvmpipe's 8-bit weights for 8-bit-UNORM textures, round-to-nearest-even UNORM stores.
 *
 * Pinned against llvmpipe goldens (tests/golden/, produced by oracle/gl_backend.c running the
 * reference's own GLSL). It is the on-box checker on the GPU machine, where llvmpipe + /root/reference
 * are absent. It is built for clarity, not speed; OpenMP over rows only so that full-size frames finish.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "ref_types.h"

typedef struct v4 {
    float x, y, z, w;
} v4;
typedef struct v3 {
    float x, y, z;
} v3;

typedef struct cpu_tex {
    int   w, h, stride, bits;
    void* data; /* memory order as uploaded: stride 3 = B,G,R ; stride 4 = B,G,R,A (texture.cpp:31-33) */
} cpu_tex;

static int g_threads = 0;
void       cpuref_set_threads(int n) { g_threads = n; }

/* ------------------------------------------------------------------ textures */

static void* cpu_tex_create(int w, int h, int stride, int bits)
{
    cpu_tex* t = (cpu_tex*)calloc(1, sizeof *t);
    t->w = w, t->h = h, t->stride = stride, t->bits = bits;
    t->data = calloc((size_t)w * h * stride, (size_t)bits / 8); /* cleared: device.cpp:207-209 */
    return t;
}
static void cpu_tex_upload(void* tex, const void* data)
{
    cpu_tex* t = (cpu_tex*)tex;
    memcpy(t->data, data, (size_t)t->w * t->h * t->stride * (t->bits / 8));
}
static void cpu_tex_read(void* tex, void* dst)
{
    cpu_tex* t = (cpu_tex*)tex;
    memcpy(dst, t->data, (size_t)t->w * t->h * t->stride * (t->bits / 8));
}
static void cpu_tex_destroy(void* tex)
{
    cpu_tex* t = (cpu_tex*)tex;
    free(t->data);
    free(t);
}
static int cpu_tex_width(void* tex) { return ((cpu_tex*)tex)->w; }
static int cpu_tex_height(void* tex) { return ((cpu_tex*)tex)->h; }

/* integer texel, GL channel order (r,g,b,a); missing channels: g=b=0, a=max */
static void texel_int(const cpu_tex* t, int x, int y, int out[4])
{
    const int    maxv = t->bits == 16 ? 65535 : 255;
    int          c[4] = {0, 0, 0, maxv};
    const size_t idx  = ((size_t)y * t->w + x) * t->stride;
    for (int k = 0; k < t->stride; ++k)
        c[k] = t->bits == 16 ? ((const uint16_t*)t->data)[idx + k] : ((const uint8_t*)t->data)[idx + k];
    if (t->stride >= 3) { /* uploaded as GL_BGR / GL_BGRA: memory B,G,R(,A) */
        out[0] = c[2], out[1] = c[1], out[2] = c[0], out[3] = c[3];
    } else {
        out[0] = c[0], out[1] = c[1], out[2] = 0, out[3] = maxv;
        if (t->stride == 1)
            out[1] = 0;
    }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Test switch (tests/parity16.py): a constant (r,g,b,a) added to the result of every fetch from a 16-bit texture
 * (source planes, keys, temporaries, the background), far below one LSB16 (1e-6 is 0.07 LSB16). It stands for ANY
 * neighbour contribution of the size a float filter weight's last bits carry, whatever the traversal order: a pixel
 * whose output moves by whole LSBs under it is one whose colour is decided by that noise (a blend mode dividing by an
 * alpha that is zero but for the noise, shader.frag:427-428), not by the inputs. */
static float g_fetch_delta[4] = {0, 0, 0, 0};
void cpuref_set_fetch_delta(float r, float g, float b, float a)
{
    g_fetch_delta[0] = r, g_fetch_delta[1] = g, g_fetch_delta[2] = b, g_fetch_delta[3] = a;
}

/* Test switch (tests/parity16.py): 16-bit filter weights within 5e-4 of a multiple of 1/8 take that value (0 for a 1:1
 * fetch, 1/4 and 3/4 for the half-width chroma planes of a 1:1 4:2:x layer), i.e. the float noise a weight carries
 * for such mappings (from a few 1e-6 at 100 pixels width to 1.2e-4 at 3840) is removed. Comparing both settings shows the pixels of a scene whose
 * output is decided by that noise. Never on in a parity comparison. */
static int g_snap_fetch = 0;
void       cpuref_set_snap_fetch(int on) { g_snap_fetch = on; }

/*
 * texture(sampler2D, coords): GL_LINEAR, CLAMP_TO_EDGE, single level.
 * 8-bit UNORM textures follow llvmpipe's fixed-point path: coordinate -> 24.8 fixed point, minus half a
 * texel, 8-bit weights, rounded integer lerps (v0 + ((w*(v1-v0) + 128) >> 8)), x first then y, each
 * intermediate an 8-bit integer (measured on Mesa 23.2.1: this model reproduces llvmpipe bit-for-bit on
 * 2x2 -> 1024x1024 magnification probes).
 * 16-bit UNORM textures follow its float path: fp32 weights, v0 + w*(v1-v0).
 */
static v4 sample(const cpu_tex* t, float s, float tt)
{
    v4 r;
    if (t->bits == 8) {
        float fu = s * (float)t->w;
        float fv = tt * (float)t->h;
        fu       = fu * 256.0f;
        fv       = fv * 256.0f;
        int iu   = (int)lrintf(fu); /* round to nearest */
        int iv   = (int)lrintf(fv);
        iu -= 128;
        iv -= 128;
        int x0 = iu >> 8, y0 = iv >> 8; /* arithmetic shift = floor */
        int wx = iu & 0xff, wy = iv & 0xff;
        int x1 = clampi(x0 + 1, 0, t->w - 1), y1 = clampi(y0 + 1, 0, t->h - 1);
        x0 = clampi(x0, 0, t->w - 1);
        y0 = clampi(y0, 0, t->h - 1);
        int c00[4], c01[4], c10[4], c11[4], o[4];
        texel_int(t, x0, y0, c00);
        texel_int(t, x1, y0, c01);
        texel_int(t, x0, y1, c10);
        texel_int(t, x1, y1, c11);
        for (int k = 0; k < 4; ++k) {
            int a = c00[k] + ((wx * (c01[k] - c00[k]) + 128) >> 8);
            int b = c10[k] + ((wx * (c11[k] - c10[k]) + 128) >> 8);
            o[k]  = a + ((wy * (b - a) + 128) >> 8);
        }
        const float sc = 1.0f / 255.0f;
        r.x = (float)o[0] * sc, r.y = (float)o[1] * sc, r.z = (float)o[2] * sc, r.w = (float)o[3] * sc;
    } else {
        float u  = s * (float)t->w - 0.5f;
        float v  = tt * (float)t->h - 0.5f;
        float fx = floorf(u), fy = floorf(v);
        float wx = u - fx, wy = v - fy;
        if (g_snap_fetch) {
            const float sx = rintf(wx * 8.0f) * 0.125f, sy = rintf(wy * 8.0f) * 0.125f;
            if (fabsf(wx - sx) < 5e-4f) wx = sx;
            if (fabsf(wy - sy) < 5e-4f) wy = sy;
            if (wx == 1.0f) fx += 1.0f, wx = 0.0f;
            if (wy == 1.0f) fy += 1.0f, wy = 0.0f;
        }
        int   x0 = (int)fx, y0 = (int)fy;
        int   x1 = clampi(x0 + 1, 0, t->w - 1), y1 = clampi(y0 + 1, 0, t->h - 1);
        x0 = clampi(x0, 0, t->w - 1);
        y0 = clampi(y0, 0, t->h - 1);
        int c00[4], c01[4], c10[4], c11[4];
        texel_int(t, x0, y0, c00);
        texel_int(t, x1, y0, c01);
        texel_int(t, x0, y1, c10);
        texel_int(t, x1, y1, c11);
        const float sc = 1.0f / 65535.0f;
        float       o[4];
        for (int k = 0; k < 4; ++k) {
            float f00 = (float)c00[k] * sc, f01 = (float)c01[k] * sc, f10 = (float)c10[k] * sc,
                  f11 = (float)c11[k] * sc;
            float a = f00 + wx * (f01 - f00);
            float b = f10 + wx * (f11 - f10);
            o[k]    = a + wy * (b - a);
        }
        r.x = o[0] + g_fetch_delta[0], r.y = o[1] + g_fetch_delta[1], r.z = o[2] + g_fetch_delta[2];
        r.w = o[3] + g_fetch_delta[3];
    }
    return r;
}

/*
 * texture(background, TexCoord2.st): the draw reads the target it is writing (glTextureBarrier, image_kernel.cpp:
 * 405-411), at the fragment's own pixel centre. On the 8-bit fixed-point path the weights quantise to exactly 0, so
 * this is the pixel's own texel. On the float path (16-bit targets) the interpolated coordinate is a few ulp off the
 * centre, the weight of a NEIGHBOUR is ~1e-6 instead of 0, and that neighbour may already hold this draw's output:
 * which neighbours do depends on the rasteriser's traversal order (llvmpipe: 4x4 blocks inside 64x64 tiles; a
 * scanline loop like cpu_draw: always the left one), i.e. it is the undefined part of reading one's own target.
 * The restatement takes the defined part: the pixel's own texel. (Found by tests/test_fuzz.py: with the bilinear
 * fetch this file was 1 LSB16 off llvmpipe AND the HIP mixer on ~0.4 % of samples of a plain stretched layer.)
 */
static int g_bg_bilinear = 0;
/* Test switch (tests/parity16.py, tests/test_oracle_golden.py): 1 = fetch the background with the bilinear float
 * filter in scanline order, i.e. mix ~1e-6 of the left / upper neighbours this draw has just written into a 16-bit
 * background. That is one of the possible outcomes of the undefined read; comparing both settings with llvmpipe
 * shows which pixels of a scene depend on it. */
void cpuref_set_background_bilinear(int on) { g_bg_bilinear = on; }

static v4 sample_own_pixel(const cpu_tex* t, float s, float tt)
{
    if (t->bits == 8 || g_bg_bilinear)
        return sample(t, s, tt);
    int x = clampi((int)floorf(s * (float)t->w), 0, t->w - 1);
    int y = clampi((int)floorf(tt * (float)t->h), 0, t->h - 1);
    int c[4];
    texel_int(t, x, y, c);
    const float sc = 1.0f / 65535.0f;
    return (v4){(float)c[0] * sc + g_fetch_delta[0], (float)c[1] * sc + g_fetch_delta[1],
                (float)c[2] * sc + g_fetch_delta[2], (float)c[3] * sc + g_fetch_delta[3]};
}

/* ------------------------------------------------------------------ GLSL helpers */

static float fminf_(float a, float b) { return a < b ? a : b; }
static float fmaxf_(float a, float b) { return a > b ? a : b; }
static float clampf(float x, float lo, float hi) { return fminf_(fmaxf_(x, lo), hi); }
static float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static float fractf(float x) { return x - floorf(x); }
static float smoothstepf(float e0, float e1, float x)
{
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

/* shader.frag:48-70 */
static v3 ContrastSaturationBrightness(v4 color, float brt, float sat, float con, int is_hd)
{
    v3 Lum = is_hd ? (v3){0.0722f, 0.7152f, 0.2126f} : (v3){0.114f, 0.587f, 0.299f};
    if (color.w > 0.0f) {
        color.x /= color.w;
        color.y /= color.w;
        color.z /= color.w;
    }
    v3    b         = {color.x * brt, color.y * brt, color.z * brt};
    float intensity = b.x * Lum.x + b.y * Lum.y + b.z * Lum.z;
    v3    s         = {mixf(intensity, b.x, sat), mixf(intensity, b.y, sat), mixf(intensity, b.z, sat)};
    v3    c         = {mixf(0.5f, s.x, con), mixf(0.5f, s.y, con), mixf(0.5f, s.z, con)};
    c.x *= color.w;
    c.y *= color.w;
    c.z *= color.w;
    return c;
}

/* Sensitivity analysis only (tests/test_gpu_parity.py): GLSL leaves pow's last bits to the implementation (llvmpipe
 * evaluates a polynomial exp2(log2(x) * y), gfx950 v_exp_f32(y * v_log_f32(x)), this file libm's powf). This knob moves
 * every pow result of the restatement by n ulps; a sample whose final value changes under it is one where the scene
 * amplifies those bits (a blend mode or chroma key comparing against a threshold). Never on in a parity comparison. */
static int g_pow_ulps = 0;
void       cpuref_set_pow_ulps(int n) { g_pow_ulps = n; }

/* shader.frag:76-86 */
static float levels1(float c, float min_in, float gamma, float max_in, float min_out, float max_out)
{
    float r = fminf_(fmaxf_(c - min_in, 0.0f) / (max_in - min_in), 1.0f);
    r       = powf(r, 1.0f / gamma);
    for (int i = 0; i < abs(g_pow_ulps); ++i)
        r = nextafterf(r, g_pow_ulps > 0 ? 2.0f : -1.0f);
    return mixf(min_out, max_out, r);
}

/* shader.frag:114-154 */
static v3 RGBToHSL(v3 color)
{
    v3    hsl  = {0, 0, 0};
    float fmin = fminf_(fminf_(color.x, color.y), color.z);
    float fmax = fmaxf_(fmaxf_(color.x, color.y), color.z);
    float delta = fmax - fmin;
    hsl.z       = (fmax + fmin) / 2.0f;
    if (delta == 0.0f) {
        hsl.x = 0.0f;
        hsl.y = 0.0f;
    } else {
        if (hsl.z < 0.5f)
            hsl.y = delta / (fmax + fmin);
        else
            hsl.y = delta / (2.0f - fmax - fmin);
        float deltaR = (((fmax - color.x) / 6.0f) + (delta / 2.0f)) / delta;
        float deltaG = (((fmax - color.y) / 6.0f) + (delta / 2.0f)) / delta;
        float deltaB = (((fmax - color.z) / 6.0f) + (delta / 2.0f)) / delta;
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

/* shader.frag:156-172 */
static float HueToRGB(float f1, float f2, float hue)
{
    if (hue < 0.0f)
        hue += 1.0f;
    else if (hue > 1.0f)
        hue -= 1.0f;
    float res;
    if ((6.0f * hue) < 1.0f)
        res = f1 + (f2 - f1) * 6.0f * hue;
    else if ((2.0f * hue) < 1.0f)
        res = f2;
    else if ((3.0f * hue) < 2.0f)
        res = f1 + (f2 - f1) * ((2.0f / 3.0f) - hue) * 6.0f;
    else
        res = f1;
    return res;
}

/* shader.frag:174-197 */
static v3 HSLToRGB(v3 hsl)
{
    v3 rgb;
    if (hsl.y == 0.0f) {
        rgb.x = rgb.y = rgb.z = hsl.z;
    } else {
        float f2;
        if (hsl.z < 0.5f)
            f2 = hsl.z * (1.0f + hsl.y);
        else
            f2 = (hsl.z + hsl.y) - (hsl.y * hsl.z);
        float f1 = 2.0f * hsl.z - f2;
        rgb.x    = HueToRGB(f1, f2, hsl.x + (1.0f / 3.0f));
        rgb.y    = HueToRGB(f1, f2, hsl.x);
        rgb.z    = HueToRGB(f1, f2, hsl.x - (1.0f / 3.0f));
    }
    return rgb;
}

/* shader.frag:208-223 scalar blend functions */
static float BlendAddf(float base, float blend) { return fminf_(base + blend, 1.0f); }
static float BlendSubstractf(float base, float blend) { return fmaxf_(base + blend - 1.0f, 0.0f); }
static float BlendLightenf(float base, float blend) { return fmaxf_(blend, base); }
static float BlendDarkenf(float base, float blend) { return fminf_(blend, base); }
static float BlendLinearLightf(float base, float blend)
{
    return blend < 0.5f ? BlendSubstractf(base, (2.0f * blend)) : BlendAddf(base, (2.0f * (blend - 0.5f)));
}
static float BlendScreenf(float base, float blend) { return 1.0f - ((1.0f - base) * (1.0f - blend)); }
static float BlendOverlayf(float base, float blend)
{
    return base < 0.5f ? (2.0f * base * blend) : (1.0f - 2.0f * (1.0f - base) * (1.0f - blend));
}
static float BlendColorDodgef(float base, float blend)
{
    return (blend == 1.0f) ? blend : fminf_(base / (1.0f - blend), 1.0f);
}
static float BlendColorBurnf(float base, float blend)
{
    return (blend == 0.0f) ? blend : fmaxf_((1.0f - ((1.0f - base) / blend)), 0.0f);
}
static float BlendVividLightf(float base, float blend)
{
    return (blend < 0.5f) ? BlendColorBurnf(base, (2.0f * blend)) : BlendColorDodgef(base, (2.0f * (blend - 0.5f)));
}
static float BlendPinLightf(float base, float blend)
{
    return (blend < 0.5f) ? BlendDarkenf(base, (2.0f * blend)) : BlendLightenf(base, (2.0f * (blend - 0.5f)));
}
static float BlendHardMixf(float base, float blend) { return (BlendVividLightf(base, blend) < 0.5f) ? 0.0f : 1.0f; }
static float BlendReflectf(float base, float blend)
{
    return (blend == 1.0f) ? blend : fminf_(base * base / (1.0f - blend), 1.0f);
}

#define PER_CH(expr_x, expr_y, expr_z) ((v3){(expr_x), (expr_y), (expr_z)})
#define BLEND3(f) PER_CH(f(b.x, s.x), f(b.y, s.y), f(b.z, s.z))

/* shader.frag:260-282 */
static v3 BlendHue(v3 base, v3 blend)
{
    v3 baseHSL = RGBToHSL(base);
    return HSLToRGB((v3){RGBToHSL(blend).x, baseHSL.y, baseHSL.z});
}
static v3 BlendSaturation(v3 base, v3 blend)
{
    v3 baseHSL = RGBToHSL(base);
    return HSLToRGB((v3){baseHSL.x, RGBToHSL(blend).y, baseHSL.z});
}
static v3 BlendColor(v3 base, v3 blend)
{
    v3 blendHSL = RGBToHSL(blend);
    return HSLToRGB((v3){blendHSL.x, blendHSL.y, RGBToHSL(base).z});
}
static v3 BlendLuminosity(v3 base, v3 blend)
{
    v3 baseHSL = RGBToHSL(base);
    return HSLToRGB((v3){baseHSL.x, baseHSL.y, RGBToHSL(blend).z});
}

/* shader.frag:387-422. b = back ("base"), s = fore ("blend"). */
static v3 get_blend_color(int mode, v3 b, v3 s)
{
    switch (mode) {
        case 0: return s;
        case 1: return BLEND3(BlendLightenf);
        case 2: return BLEND3(BlendDarkenf);
        case 3: return PER_CH(b.x * s.x, b.y * s.y, b.z * s.z);
        case 4: return PER_CH((b.x + s.x) / 2.0f, (b.y + s.y) / 2.0f, (b.z + s.z) / 2.0f);
        case 5: return BLEND3(BlendAddf);
        case 6: return BLEND3(BlendSubstractf);
        case 7: return PER_CH(fabsf(b.x - s.x), fabsf(b.y - s.y), fabsf(b.z - s.z));
        case 8:
            return PER_CH(1.0f - fabsf(1.0f - b.x - s.x), 1.0f - fabsf(1.0f - b.y - s.y),
                          1.0f - fabsf(1.0f - b.z - s.z));
        case 9:
            return PER_CH(b.x + s.x - 2.0f * b.x * s.x, b.y + s.y - 2.0f * b.y * s.y, b.z + s.z - 2.0f * b.z * s.z);
        case 10: return BLEND3(BlendScreenf);
        case 11: return BLEND3(BlendOverlayf);
        /* case 12 (soft light) is commented out in the reference -> falls to normal */
        case 13: return PER_CH(BlendOverlayf(s.x, b.x), BlendOverlayf(s.y, b.y), BlendOverlayf(s.z, b.z));
        case 14: return BLEND3(BlendColorDodgef);
        case 15: return BLEND3(BlendColorBurnf);
        case 16: return BLEND3(BlendAddf);
        case 17: return BLEND3(BlendSubstractf);
        case 18: return BLEND3(BlendLinearLightf);
        case 19: return BLEND3(BlendVividLightf);
        case 20: return BLEND3(BlendPinLightf);
        case 21: return BLEND3(BlendHardMixf);
        case 22: return BLEND3(BlendReflectf);
        case 23: return PER_CH(BlendReflectf(s.x, b.x), BlendReflectf(s.y, b.y), BlendReflectf(s.z, b.z));
        case 24:
            return PER_CH(fminf_(b.x, s.x) - fmaxf_(b.x, s.x) + 1.0f, fminf_(b.y, s.y) - fmaxf_(b.y, s.y) + 1.0f,
                          fminf_(b.z, s.z) - fmaxf_(b.z, s.z) + 1.0f);
        case 25: return BlendHue(b, s);
        case 26: return BlendSaturation(b, s);
        case 27: return BlendColor(b, s);
        case 28: return BlendLuminosity(b, s);
    }
    return s;
}

/* shader.frag:304-321 */
static v3 rgb2hsv(v3 c)
{
    /* K = (0, -1/3, 2/3, -1) */
    float st = stepf(c.z, c.y); /* step(c.b, c.g) */
    v4    p  = {mixf(c.z, c.y, st), mixf(c.y, c.z, st), mixf(-1.0f, 0.0f, st), mixf(2.0f / 3.0f, -1.0f / 3.0f, st)};
    float s2 = stepf(p.x, c.x);
    v4    q  = {mixf(p.x, c.x, s2), mixf(p.y, p.y, s2), mixf(p.w, p.z, s2), mixf(c.x, p.x, s2)};
    float d  = q.x - fminf_(q.w, q.y);
    float e  = 1.0e-10f;
    v3    r  = {fabsf(q.z + (q.w - q.y) / (6.0f * d + e)), d / (q.x + e), q.x};
    return r;
}
static v3 hsv2rgb(v3 c)
{
    /* K = (1, 2/3, 1/3, 3) */
    v3 p = {fabsf(fractf(c.x + 1.0f) * 6.0f - 3.0f), fabsf(fractf(c.x + 2.0f / 3.0f) * 6.0f - 3.0f),
            fabsf(fractf(c.x + 1.0f / 3.0f) * 6.0f - 3.0f)};
    v3 r = {c.z * mixf(1.0f, clampf(p.x - 1.0f, 0.0f, 1.0f), c.y), c.z * mixf(1.0f, clampf(p.y - 1.0f, 0.0f, 1.0f), c.y),
            c.z * mixf(1.0f, clampf(p.z - 1.0f, 0.0f, 1.0f), c.y)};
    return r;
}

/* shader.frag:323-383; input/output in real (r,g,b,a) order */
static v4 ChromaOnCustomColor(v4 c, const ref_draw* d)
{
    v3 hsv = rgb2hsv((v3){c.x, c.y, c.z});
    /* ColorDistance :342-352 */
    float hueDiff = (0.5f - fabsf(fabsf(hsv.x - d->chroma_target_hue) - 0.5f)) * 2.0f;
    float satDiff = fminf_(0.0f, d->chroma_min_saturation - hsv.y);
    float briDiff = fminf_(0.0f, d->chroma_min_brightness - hsv.z);
    float sbScore = fmaxf_(briDiff, satDiff);
    float hueScore = hueDiff - d->chroma_hue_width;
    float distance = -hueScore * sbScore;
    float dd       = distance * -2.0f + 1.0f;
    /* supress_spill :354-369 */
    v3    sp   = hsv;
    float diff = hsv.x - d->chroma_target_hue;
    diff       = diff < -0.5f ? diff + 1.0f : (diff > 0.5f ? diff - 1.0f : diff);
    float dist = fabsf(diff) / d->chroma_spill_suppress;
    if (dist < 1.0f) {
        sp.x = diff < 0.0f ? d->chroma_target_hue - d->chroma_spill_suppress
                           : d->chroma_target_hue + d->chroma_spill_suppress;
        sp.y *= fminf_(1.0f, dist + d->chroma_spill_suppress_saturation);
    }
    v3    rgb   = hsv2rgb(sp);
    float alpha = 1.0f - smoothstepf(1.0f, d->chroma_softness, dd);
    v4    sup   = {rgb.x * alpha, rgb.y * alpha, rgb.z * alpha, 1.0f * alpha};
    if (d->chroma_show_mask)
        return (v4){sup.w, sup.w, sup.w, 1.0f};
    return sup;
}

/* shader.frag:441-467; returns internal order x=blue(Cb term) y=green z=red(Cr term) w=a */
static v4 ycbcra_to_rgba(float Y, float Cb, float Cr, float A, int is_hd)
{
    v4 c;
    if (is_hd) {
        c.z = (1.164f * (Y * 255.0f - 16.0f) + 1.793f * (Cr * 255.0f - 128.0f)) / 255.0f;
        c.y = (1.164f * (Y * 255.0f - 16.0f) - 0.534f * (Cr * 255.0f - 128.0f) - 0.213f * (Cb * 255.0f - 128.0f)) /
              255.0f;
        c.x = (1.164f * (Y * 255.0f - 16.0f) + 2.115f * (Cb * 255.0f - 128.0f)) / 255.0f;
    } else {
        c.z = (1.164f * (Y * 255.0f - 16.0f) + 1.596f * (Cr * 255.0f - 128.0f)) / 255.0f;
        c.y = (1.164f * (Y * 255.0f - 16.0f) - 0.813f * (Cr * 255.0f - 128.0f) - 0.391f * (Cb * 255.0f - 128.0f)) /
              255.0f;
        c.x = (1.164f * (Y * 255.0f - 16.0f) + 2.018f * (Cb * 255.0f - 128.0f)) / 255.0f;
    }
    c.w = A;
    return c;
}

/* shader.frag:474-521. Returns internal order: (.r,.g,.b,.a) slots of the GLSL vec4 = (x,y,z,w). */
static v4 get_rgba_color(const ref_draw* d, float s, float t)
{
    const cpu_tex* const* pl = (const cpu_tex* const*)d->planes;
    v4                    a;
    switch (d->pixel_format) {
        case 0: a = sample(pl[0], s, t); return (v4){a.x, a.x, a.x, 1.0f};
        case 1: a = sample(pl[0], s, t); return (v4){a.z, a.y, a.x, a.w}; /* .bgra */
        case 2: a = sample(pl[0], s, t); return a;                        /* .rgba */
        case 3: a = sample(pl[0], s, t); return (v4){a.w, a.x, a.y, a.z}; /* .argb */
        case 4: a = sample(pl[0], s, t); return (v4){a.y, a.z, a.w, a.x}; /* .gbar */
        case 5: {
            float y  = sample(pl[0], s, t).x;
            float cb = sample(pl[1], s, t).x;
            float cr = sample(pl[2], s, t).x;
            return ycbcra_to_rgba(y, cb, cr, 1.0f, d->is_hd);
        }
        case 6: {
            float y  = sample(pl[0], s, t).x;
            float cb = sample(pl[1], s, t).x;
            float cr = sample(pl[2], s, t).x;
            float al = sample(pl[3], s, t).x;
            return ycbcra_to_rgba(y, cb, cr, al, d->is_hd);
        }
        case 7: a = sample(pl[0], s, t); return (v4){(a.x - 0.065f) / 0.859f, (a.x - 0.065f) / 0.859f,
                                                     (a.x - 0.065f) / 0.859f, 1.0f};
        case 8: a = sample(pl[0], s, t); return (v4){a.z, a.y, a.x, 1.0f}; /* .bgr */
        case 9: a = sample(pl[0], s, t); return (v4){a.x, a.y, a.z, 1.0f}; /* .rgb */
        case 10: {
            float y  = sample(pl[0], s, t).y;
            v4    c  = sample(pl[1], s, t);
            return ycbcra_to_rgba(y, c.z, c.x, 1.0f, d->is_hd); /* cb = .b, cr = .r */
        }
    }
    return (v4){0, 0, 0, 0};
}

/* float -> UNORM store as llvmpipe does it: clamp, scale by (2^n-1)/2^n, add 2^(23-n), take the low
 * n mantissa bits (= round-to-nearest-even of x*(2^n-1)). */
static unsigned to_unorm(float x, int bits)
{
    if (!(x > 0.0f))
        x = 0.0f; /* also NaN -> 0 */
    if (x > 1.0f)
        x = 1.0f;
    union {
        float    f;
        uint32_t u;
    } cv;
    if (bits == 8) {
        cv.f = x * (255.0f / 256.0f) + 32768.0f;
        return cv.u & 0xffu;
    }
    cv.f = x * (65535.0f / 65536.0f) + 128.0f;
    return cv.u & 0xffffu;
}

/* shader.frag:523-542 main(), for one fragment. tc = TexCoord (s*q,t*q,r,q), tc2 = TexCoord2.st */
static v4 fragment(const ref_draw* d, v4 tc, float s2, float t2)
{
    float s = tc.x / tc.w, t = tc.y / tc.w;
    v4    color = get_rgba_color(d, s, t);
    if (d->chroma) { /* chroma_key :436-439: c.bgra in, .bgra out */
        v4 r  = ChromaOnCustomColor((v4){color.z, color.y, color.x, color.w}, d);
        color = (v4){r.z, r.y, r.x, r.w};
    }
    if (d->levels) {
        color.x = levels1(color.x, d->min_input, d->gamma, d->max_input, d->min_output, d->max_output);
        color.y = levels1(color.y, d->min_input, d->gamma, d->max_input, d->min_output, d->max_output);
        color.z = levels1(color.z, d->min_input, d->gamma, d->max_input, d->min_output, d->max_output);
    }
    if (d->csb) {
        v3 c    = ContrastSaturationBrightness(color, d->brt, d->sat, d->con, d->is_hd);
        color.x = c.x, color.y = c.y, color.z = c.z;
    }
    if (d->has_local_key) {
        float k = sample((const cpu_tex*)d->local_key, s2, t2).x;
        color.x *= k, color.y *= k, color.z *= k, color.w *= k;
    }
    if (d->has_layer_key) {
        float k = sample((const cpu_tex*)d->layer_key, s2, t2).x;
        color.x *= k, color.y *= k, color.z *= k, color.w *= k;
    }
    color.x *= d->opacity, color.y *= d->opacity, color.z *= d->opacity, color.w *= d->opacity;
    if (d->invert)
        color = (v4){1.0f - color.x, 1.0f - color.y, 1.0f - color.z, 1.0f - color.w};
    if (d->blend_mode >= 0) { /* blend() :424-434 */
        v4 bs   = sample_own_pixel((const cpu_tex*)d->background, s2, t2);
        v4 back = {bs.z, bs.y, bs.x, bs.w}; /* .bgra */
        v4 fore = color;
        if (d->blend_mode != 0) {
            float ba = back.w + 0.0000001f, fa = fore.w + 0.0000001f;
            v3    bc = get_blend_color(d->blend_mode, (v3){back.x / ba, back.y / ba, back.z / ba},
                                    (v3){fore.x / fa, fore.y / fa, fore.z / fa});
            fore.x = bc.x * fore.w, fore.y = bc.y * fore.w, fore.z = bc.z * fore.w;
        }
        if (d->keyer == 1)
            color = (v4){fore.x + back.x, fore.y + back.y, fore.z + back.z, fore.w + back.w};
        else {
            float ia = 1.0f - fore.w;
            color    = (v4){fore.x + ia * back.x, fore.y + ia * back.y, fore.z + ia * back.z, fore.w + ia * back.w};
        }
    }
    return (v4){color.z, color.y, color.x, color.w}; /* fragColor = color.bgra -> (R,G,B,A) */
}

/* ------------------------------------------------------------------ rasteriser */

#define SUBPIX 256 /* 8 sub-pixel bits */

typedef struct tri_setup {
    int64_t ax, ay, bx, by, cx, cy; /* snapped vertices, 24.8, pixel-centre space */
    int     valid;
    int64_t bias[3]; /* fill-rule bias per edge */
    double  fx[3], fy[3]; /* float window positions (pixel-centre space) for interpolation */
    double  attr[3][6]; /* s*q, t*q, r, q, vx, vy per vertex */
    double  inv_area;
} tri_setup;

static int64_t iround64(double v) { return (int64_t)llrint(v); }

/* edge function, positive inside after orientation fix */
static int64_t edge(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t px, int64_t py)
{
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}

static int is_top_left(int64_t ax, int64_t ay, int64_t bx, int64_t by)
{
    /* y grows downwards (memory-row space); the triangle is oriented so that inside is E > 0, i.e.
       clockwise on screen: left edges run upwards (dy < 0), the top edge runs left to right.
       Validated against llvmpipe (tests/test_oracle_golden.py::fill rule). */
    int64_t dx = bx - ax, dy = by - ay;
    return (dy < 0) || (dy == 0 && dx > 0);
}

static void setup_tri(tri_setup* ts, const float pos[3][2], const float attr[3][6], int W, int H)
{
    /* shader.vert:13-15 then viewport transform (image_kernel.cpp:309) */
    double wx[3], wy[3];
    for (int i = 0; i < 3; ++i) {
        float ndc_x = pos[i][0] * 2.0f - 1.0f;
        float ndc_y = pos[i][1] * 2.0f - 1.0f;
        float fx    = ndc_x * ((float)W * 0.5f) + ((float)W * 0.5f);
        float fy    = ndc_y * ((float)H * 0.5f) + ((float)H * 0.5f);
        wx[i]       = (double)fx - 0.5; /* pixel-centre space: pixel (x,y) centre sits at integer (x,y) */
        wy[i]       = (double)fy - 0.5;
    }
    int64_t X[3], Y[3];
    for (int i = 0; i < 3; ++i) {
        X[i] = iround64(wx[i] * SUBPIX);
        Y[i] = iround64(wy[i] * SUBPIX);
    }
    int64_t area = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
    ts->valid    = area != 0;
    if (!ts->valid)
        return;
    int o[3] = {0, 1, 2};
    if (area < 0) { /* make orientation positive (no face culling in the reference) */
        o[1] = 2;
        o[2] = 1;
    }
    ts->ax = X[o[0]], ts->ay = Y[o[0]];
    ts->bx = X[o[1]], ts->by = Y[o[1]];
    ts->cx = X[o[2]], ts->cy = Y[o[2]];
    ts->bias[0] = is_top_left(ts->ax, ts->ay, ts->bx, ts->by) ? 0 : -1;
    ts->bias[1] = is_top_left(ts->bx, ts->by, ts->cx, ts->cy) ? 0 : -1;
    ts->bias[2] = is_top_left(ts->cx, ts->cy, ts->ax, ts->ay) ? 0 : -1;
    for (int i = 0; i < 3; ++i) {
        ts->fx[i] = wx[o[i]];
        ts->fy[i] = wy[o[i]];
        for (int k = 0; k < 6; ++k)
            ts->attr[i][k] = (double)attr[o[i]][k];
    }
    double a     = (ts->fx[1] - ts->fx[0]) * (ts->fy[2] - ts->fy[0]) - (ts->fy[1] - ts->fy[0]) * (ts->fx[2] - ts->fx[0]);
    ts->inv_area = a != 0.0 ? 1.0 / a : 0.0;
}

static int covers(const tri_setup* ts, int x, int y)
{
    int64_t px = (int64_t)x * SUBPIX, py = (int64_t)y * SUBPIX;
    int64_t e0 = edge(ts->ax, ts->ay, ts->bx, ts->by, px, py) + ts->bias[0];
    int64_t e1 = edge(ts->bx, ts->by, ts->cx, ts->cy, px, py) + ts->bias[1];
    int64_t e2 = edge(ts->cx, ts->cy, ts->ax, ts->ay, px, py) + ts->bias[2];
    return e0 >= 0 && e1 >= 0 && e2 >= 0;
}

static void interpolate(const tri_setup* ts, int x, int y, float out[6])
{
    double px = (double)x, py = (double)y;
    /* barycentrics from the float positions (w = 1: perspective-correct == affine) */
    double l1 = ((px - ts->fx[0]) * (ts->fy[2] - ts->fy[0]) - (py - ts->fy[0]) * (ts->fx[2] - ts->fx[0])) * ts->inv_area;
    double l2 = ((ts->fx[1] - ts->fx[0]) * (py - ts->fy[0]) - (ts->fy[1] - ts->fy[0]) * (px - ts->fx[0])) * ts->inv_area;
    for (int k = 0; k < 6; ++k)
        out[k] = (float)(ts->attr[0][k] + l1 * (ts->attr[1][k] - ts->attr[0][k]) + l2 * (ts->attr[2][k] - ts->attr[0][k]));
}

/* Test switch (tests/parity16.py): every store into a 16-bit render target lands on the code above or below the
 * rounded one with probability 1/4 each (seeded hash of draw, pixel and channel; seeds 0xFFFFFFFF / 0xFFFFFFFE: every
 * store one code up / down). It stands for an implementation
 * whose arithmetic differs from the exact sequence by a few 1e-7 and therefore rounds to the neighbouring code now and
 * then: a pixel whose final value moves by more than 2 LSB16 under it is one where the scene amplifies one LSB16 of an
 * intermediate render target (a blend mode dividing by alpha, shader.frag:427-428). Never on in a parity comparison. */
static unsigned g_store_flip = 0; /* 0 = off, else the seed */
static unsigned g_draw_index = 0;
void            cpuref_set_store_flip(unsigned seed) { g_store_flip = seed, g_draw_index = 0; }
/* The deterministic form of the same probe (tests/parity16.py, tiers 3 and 4): the stores of ONE draw (draws are counted
 * from 1 in execution order since the last reset) land off their code at EVERY pixel, each sample of the target's texel
 * by its own delta (memory order B, G, R, A; a key target has the first one only). One draw at a time; the test walks the
 * sign patterns: colour and alpha up, down, apart and against each other (rgb / alpha of a blend mode, shader.frag:
 * 427-428, moves most when they err in opposite directions), and for the HSL modes, whose hue and saturation hang on
 * DIFFERENCES between channels, every pattern of -1 / 0 / +1 over the four samples. What one LSB16 of that draw's
 * render target is worth at every pixel of the final frame. draw 0 = off. */
static int g_flip_draw = 0, g_flip_delta[4] = {0, 0, 0, 0};
void       cpuref_set_store_flip_draw(int draw, int db, int dg, int dr, int da)
{
    g_flip_draw = draw, g_draw_index = 0;
    g_flip_delta[0] = db, g_flip_delta[1] = dg, g_flip_delta[2] = dr, g_flip_delta[3] = da;
}
void cpuref_reset_draw_count(void) { g_draw_index = 0; }
int  cpuref_draw_count(void) { return (int)g_draw_index; } /* draws executed since the last reset */
static int store_flip(int x, int y, int k)
{
    if (g_flip_draw) /* the deterministic probe */
        return (unsigned)g_flip_draw == g_draw_index ? g_flip_delta[k & 3] : 0;
    if (g_store_flip >= 0xFFFFFFFEu) /* every store one code up (..FF) / down (..FE) */
        return g_store_flip == 0xFFFFFFFFu ? 1 : -1;
    uint64_t z = ((uint64_t)g_store_flip << 48) ^ ((uint64_t)g_draw_index << 34) ^ ((uint64_t)y << 18) ^ ((uint64_t)x << 2) ^ (uint64_t)k;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const unsigned r = (unsigned)(z >> 40) & 3u;
    return r == 0 ? -1 : (r == 1 ? 1 : 0);
}

static void store(cpu_tex* bg, int x, int y, v4 rgba)
{
    const size_t idx = ((size_t)y * bg->w + x) * bg->stride;
    /* memory order of an RGBA target as read back with GL_BGRA: B,G,R,A; R8 target keeps .r */
    float ch[4];
    int   n;
    if (bg->stride == 4) {
        ch[0] = rgba.z, ch[1] = rgba.y, ch[2] = rgba.x, ch[3] = rgba.w;
        n = 4;
    } else {
        ch[0] = rgba.x;
        n     = 1;
    }
    for (int k = 0; k < n; ++k) {
        unsigned v = to_unorm(ch[k], bg->bits);
        if ((g_store_flip || g_flip_draw) && bg->bits == 16) {
            const int f = (int)v + store_flip(x, y, k);
            v           = (unsigned)(f < 0 ? 0 : (f > 65535 ? 65535 : f));
        }
        if (bg->bits == 16)
            ((uint16_t*)bg->data)[idx + k] = (uint16_t)v;
        else
            ((uint8_t*)bg->data)[idx + k] = (uint8_t)v;
    }
}

static void cpu_draw(const ref_draw* d)
{
    g_draw_index++;
    cpu_tex* bg = (cpu_tex*)d->background;
    const int W = bg->w, H = bg->h;

    /* vertex fetch narrows GL_DOUBLE attributes to float (image_kernel.cpp:399-400, shader.vert:2-3) */
    float pos[4][2], attr[4][6];
    for (int i = 0; i < 4; ++i) {
        pos[i][0]  = (float)d->coords[i].vertex_x;
        pos[i][1]  = (float)d->coords[i].vertex_y;
        attr[i][0] = (float)d->coords[i].texture_x;
        attr[i][1] = (float)d->coords[i].texture_y;
        attr[i][2] = (float)d->coords[i].texture_r;
        attr[i][3] = (float)d->coords[i].texture_q;
        attr[i][4] = pos[i][0]; /* TexCoord2 = Position (shader.vert:12) */
        attr[i][5] = pos[i][1];
    }
    static const int tri_idx[2][3] = {{0, 1, 2}, {0, 2, 3}}; /* image_kernel.cpp:383-384 */
    tri_setup        ts[2];
    int              xmin = W, xmax = -1, ymin = H, ymax = -1;
    for (int k = 0; k < 2; ++k) {
        float p[3][2], a[3][6];
        for (int i = 0; i < 3; ++i) {
            memcpy(p[i], pos[tri_idx[k][i]], sizeof p[i]);
            memcpy(a[i], attr[tri_idx[k][i]], sizeof a[i]);
        }
        setup_tri(&ts[k], p, a, W, H);
        if (!ts[k].valid)
            continue;
        int64_t xs[3] = {ts[k].ax, ts[k].bx, ts[k].cx}, ys[3] = {ts[k].ay, ts[k].by, ts[k].cy};
        for (int i = 0; i < 3; ++i) {
            int lo_x = (int)floor((double)xs[i] / SUBPIX), hi_x = (int)ceil((double)xs[i] / SUBPIX);
            int lo_y = (int)floor((double)ys[i] / SUBPIX), hi_y = (int)ceil((double)ys[i] / SUBPIX);
            if (lo_x < xmin) xmin = lo_x;
            if (hi_x > xmax) xmax = hi_x;
            if (lo_y < ymin) ymin = lo_y;
            if (hi_y > ymax) ymax = hi_y;
        }
    }
    if (xmin < 0) xmin = 0;
    if (ymin < 0) ymin = 0;
    if (xmax > W - 1) xmax = W - 1;
    if (ymax > H - 1) ymax = H - 1;
    if (d->scissor) { /* glScissor box in window coords, origin = memory row 0 (no flip for FBO textures) */
        if (d->sc_x > xmin) xmin = d->sc_x;
        if (d->sc_y > ymin) ymin = d->sc_y;
        if (d->sc_x + d->sc_w - 1 < xmax) xmax = d->sc_x + d->sc_w - 1;
        if (d->sc_y + d->sc_h - 1 < ymax) ymax = d->sc_y + d->sc_h - 1;
    }
#ifdef _OPENMP
    int nth = g_threads > 0 ? g_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 8) num_threads(nth)
#endif
    for (int y = ymin; y <= ymax; ++y) {
        if (d->field_mode == 1 && (y & 1) != 0)
            continue;
        if (d->field_mode == 2 && (y & 1) != 1)
            continue;
        for (int x = xmin; x <= xmax; ++x) {
            int k = -1;
            if (ts[0].valid && covers(&ts[0], x, y))
                k = 0;
            else if (ts[1].valid && covers(&ts[1], x, y))
                k = 1;
            if (k < 0)
                continue;
            float a[6];
            interpolate(&ts[k], x, y, a);
            v4 out = fragment(d, (v4){a[0], a[1], a[2], a[3]}, a[4], a[5]);
            store(bg, x, y, out);
        }
    }
}

static const ref_backend g_cpu_backend = {cpu_tex_create, cpu_tex_upload, cpu_tex_read, cpu_tex_destroy,
                                          cpu_tex_width,  cpu_tex_height, cpu_draw};

const ref_backend* cpuref_backend(void) { return &g_cpu_backend; }
