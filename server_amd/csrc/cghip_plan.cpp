/*
 * cghip_plan.cpp -- host planner: reference draw calls -> KOp program for the fused kernel.
 *
 * Restates, for the HIP path, the host part of image_kernel::draw
 * (accelerator/ogl/image/image_kernel.cpp:134-375: early-outs, crop / perspective / rotate / translate of
 * the quad, off-screen cull, uniforms, scissor, projective q) and the fixed-function stage that follows it
 * on a GL device (viewport transform, 24.8 sub-pixel snapping, two triangles (0,1,2),(0,2,3) with the
 * top-left fill rule, per-triangle affine varyings). Also maps the renderer's temporaries
 * (image_mixer.cpp:128,177,187) onto the kernel's register surfaces by liveness.
 * Pure host code: no HIP calls, unit-testable without a GPU.
 */
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>

#include "cghip_internal.h"

namespace cghip {

namespace {

struct vtx {
    double vx, vy, s, t, r, q;
};

bool coords_equal(const cghip_coord* a, const double ref[4][4])
{ /* frame_geometry::coord::operator== over a quad (geometry.cpp:35-39) */
    for (int i = 0; i < 4; ++i)
        if (!(a[i].vertex_x == ref[i][0] && a[i].vertex_y == ref[i][1] && a[i].texture_x == ref[i][2] &&
              a[i].texture_y == ref[i][3] && a[i].texture_r == 0.0 && a[i].texture_q == 1.0))
            return false;
    return true;
}

const double kDefault[4][4]      = {{0, 0, 0, 0}, {1, 0, 1, 0}, {1, 1, 1, 1}, {0, 1, 0, 1}}; /* geometry.cpp:64-76 */
const double kDefaultVflip[4][4] = {{0, 0, 0, 1}, {1, 0, 1, 1}, {1, 1, 1, 0}, {0, 1, 0, 0}}; /* geometry.cpp:77-89 */

bool line_intersection(double p0x, double p0y, double p1x, double p1y, double p2x, double p2y, double p3x,
                       double p3y, double& rx, double& ry)
{ /* image_kernel.cpp:46-74 */
    const double s1x = p1x - p0x, s1y = p1y - p0y, s2x = p3x - p2x, s2y = p3y - p2y;
    const double den = -s2x * s1y + s1x * s2y;
    const double s   = (-s1y * (p0x - p2x) + s1x * (p0y - p2y)) / den;
    const double t   = (s2x * (p0y - p2y) - s2y * (p0x - p2x)) / den;
    if (s >= 0 && s <= 1 && t >= 0 && t <= 1) {
        rx = p0x + t * s1x;
        ry = p0y + t * s1y;
        return true;
    }
    return false;
}

double hyp(double x1, double y1, double x2, double y2) { return std::sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }

/* Quad after image_kernel::draw's vertex math; false = the reference returns before drawing. */
bool transform_quad(const cghip_draw& d, vtx out[4])
{
    const cghip_transform& t = d.transform;
    for (int i = 0; i < 4; ++i)
        out[i] = vtx{d.coords[i].vertex_x,  d.coords[i].vertex_y,  d.coords[i].texture_x,
                     d.coords[i].texture_y, d.coords[i].texture_r, d.coords[i].texture_q};
    const bool is_default = coords_equal(d.coords, kDefault) || coords_equal(d.coords, kDefaultVflip); /* :158-159 */

    const double pers[4][2] = {{t.pers_ul[0], t.pers_ul[1]},
                               {t.pers_ur[0] - 1.0, t.pers_ur[1]},
                               {t.pers_lr[0] - 1.0, t.pers_lr[1] - 1.0},
                               {t.pers_ll[0], t.pers_ll[1] - 1.0}}; /* :164-169 */
    const double aspect = d.aspect_ratio, ca = std::cos(t.angle), sa = std::sin(t.angle);
    for (int i = 0; i < 4; ++i) {
        vtx& c = out[i];
        if (is_default) {
            c.vx = std::min(std::max(c.vx, t.crop_ul[0]), t.crop_lr[0]); /* do_crop :171-185 */
            c.vy = std::min(std::max(c.vy, t.crop_ul[1]), t.crop_lr[1]);
            c.s  = std::min(std::max(c.s, t.crop_ul[0]), t.crop_lr[0]);
            c.t  = std::min(std::max(c.t, t.crop_ul[1]), t.crop_lr[1]);
            c.vx += pers[i][0]; /* do_perspective :186-194 */
            c.vy += pers[i][1];
        }
        const double ox = (c.vx - t.anchor[0]) * t.fill_scale[0]; /* rotate :195-201 */
        const double oy = (c.vy - t.anchor[1]) * t.fill_scale[1] / aspect;
        c.vx            = ox * ca - oy * sa;
        c.vy            = (ox * sa + oy * ca) * aspect;
        c.vx += t.fill_translation[0]; /* move :202-205 */
        c.vy += t.fill_translation[1];
    }
    bool l = true, r = true, a = true, b = true; /* is_outside_screen :97-107 */
    for (int i = 0; i < 4; ++i) {
        l = l && out[i].vx < 0.0;
        r = r && out[i].vx > 1.0;
        a = a && out[i].vy < 0.0;
        b = b && out[i].vy > 1.0;
    }
    if (l || r || a || b)
        return false;

    /* projective q :335-375 */
    const double *cu = t.crop_ul, *cl = t.crop_lr;
    double        ix, iy;
    if (line_intersection(pers[0][0] + cu[0], pers[0][1] + cu[1], pers[2][0] + cl[0], pers[2][1] + cl[1],
                          pers[1][0] + cl[0], pers[1][1] + cu[1], pers[3][0] + cu[0], pers[3][1] + cl[1], ix, iy) &&
        is_default) {
        const double d0   = hyp(pers[3][0] + cu[0], pers[3][1] + cl[1], ix, iy);
        const double d1   = hyp(pers[2][0] + cl[0], pers[2][1] + cl[1], ix, iy);
        const double d2   = hyp(pers[1][0] + cl[0], pers[1][1] + cu[1], ix, iy);
        const double d3   = hyp(pers[0][0] + cu[0], pers[0][1] + cu[1], ix, iy);
        const double q[4] = {(d3 + d1) / d1, (d2 + d0) / d0, (d1 + d3) / d3, (d0 + d2) / d2};
        for (int i = 0; i < 4; ++i) {
            out[i].q = q[i];
            out[i].s *= q[i];
            out[i].t *= q[i];
        }
    }
    return true;
}

struct raster_tri {
    int64_t X[3], Y[3];
    double  fx[3], fy[3], attr[3][3];
    bool    valid;
};

/* vertex stage (shader.vert:8-16) + viewport + snapping for one triangle */
raster_tri setup_tri(const vtx q[4], const int idx[3], int W, int H)
{
    raster_tri tr{};
    for (int i = 0; i < 3; ++i) {
        const vtx&  v  = q[idx[i]];
        const float px = (float)v.vx, py = (float)v.vy; /* GL_DOUBLE attribs narrowed to float */
        const float nx = px * 2.0f - 1.0f, ny = py * 2.0f - 1.0f;
        const float wx = nx * ((float)W * 0.5f) + ((float)W * 0.5f);
        const float wy = ny * ((float)H * 0.5f) + ((float)H * 0.5f);
        tr.fx[i]       = (double)wx - 0.5; /* pixel-centre space */
        tr.fy[i]       = (double)wy - 0.5;
        tr.X[i]        = (int64_t)std::llrint(tr.fx[i] * 256.0);
        tr.Y[i]        = (int64_t)std::llrint(tr.fy[i] * 256.0);
        tr.attr[i][0]  = (double)(float)v.s;
        tr.attr[i][1]  = (double)(float)v.t;
        tr.attr[i][2]  = (double)(float)v.q;
    }
    const int64_t area = (tr.X[1] - tr.X[0]) * (tr.Y[2] - tr.Y[0]) - (tr.Y[1] - tr.Y[0]) * (tr.X[2] - tr.X[0]);
    tr.valid           = area != 0;
    if (area < 0) { /* no face culling in the reference: orient clockwise-on-screen */
        std::swap(tr.X[1], tr.X[2]);
        std::swap(tr.Y[1], tr.Y[2]);
        std::swap(tr.fx[1], tr.fx[2]);
        std::swap(tr.fy[1], tr.fy[2]);
        for (int k = 0; k < 3; ++k)
            std::swap(tr.attr[1][k], tr.attr[2][k]);
    }
    return tr;
}

void emit_tri(const raster_tri& tr, KTri& k)
{
    for (int e = 0; e < 3; ++e) {
        const int     a = e, b = (e + 1) % 3;
        const int64_t dx = tr.X[b] - tr.X[a], dy = tr.Y[b] - tr.Y[a];
        /* E(px,py) = dx*(py-ay) - dy*(px-ax) with px = 256x, py = 256y; top-left: ties belong to edges that
           run upwards (left) or left-to-right horizontally (top) */
        const bool    top_left = (dy < 0) || (dy == 0 && dx > 0);
        const int64_t c        = -dx * tr.Y[a] + dy * tr.X[a] + (top_left ? 0 : -1);
        k.e[e][0]              = (double)(-dy * 256);
        k.e[e][1]              = (double)(dx * 256);
        k.e[e][2]              = (double)c;
    }
    /* varyings: plane through the three vertices, in pixel-index space */
    const double x0 = tr.fx[0], y0 = tr.fy[0];
    const double ax = tr.fx[1] - x0, ay = tr.fy[1] - y0, bx = tr.fx[2] - x0, by = tr.fy[2] - y0;
    const double det = ax * by - ay * bx;
    const double inv = det != 0.0 ? 1.0 / det : 0.0;
    for (int j = 0; j < 3; ++j) {
        const double d1 = tr.attr[1][j] - tr.attr[0][j], d2 = tr.attr[2][j] - tr.attr[0][j];
        const double gx = (d1 * by - d2 * ay) * inv;
        const double gy = (d2 * ax - d1 * bx) * inv;
        k.a[j][1]       = gx;
        k.a[j][2]       = gy;
        k.a[j][0]       = tr.attr[0][j] - gx * x0 - gy * y0;
    }
}

int ceil_div256(int64_t v) { return (int)((v >= 0) ? (v + 255) / 256 : -((-v) / 256)); }
int floor_div256(int64_t v) { return (int)((v >= 0) ? v / 256 : -((-v + 255) / 256)); }

} // namespace

/* Which draws the 4-pixels-per-lane kernel (cghip_composite_fast) can run: unit-scale sources whose texels
 * can be fetched with aligned vector loads, normal blend into the frame or a key surface. */
static void classify_fast_draw(KOp& op, int frame_w);
static int  g_no_sampled_fast = 0; /* test hook: sampled draws stay with the general kernel */
static void classify_fast(KOp& op, int frame_w)
{
    classify_fast_draw(op, frame_w);
    /* Axis-aligned boxes at any scale (picture-in-picture) and 1:1 sources off the 4-texel grid of the aligned bodies,
       from planes the general kernel has a straight-line sampling body for (classify_sampling, called before):
       the lane fetches four bilinear samples - source rows and row weight are wave-uniform - and zeroes the pixels the
       box does not cover, which leaves the target as it is unless the draw inverts. Rotated and perspective quads stay
       with the 1-pixel-per-lane kernel: its per-pixel colour inlined four times was measured at 10.9 (rotated) and 7.1
       (perspective) ps per pixel against 6.3 and 5.4 there (tests/gpu_probe.py, 16 layers 2160p). */
    if (op.fast_class == FC_NONE && op.kind == KOP_DRAW_PLANES && op.blend_mode == 0 && !(op.flags & KF_INVERT) &&
        !g_no_sampled_fast && ((op.flags & KF_AXIS) || (op.flags & KF_UNIT_SCALE)) &&
        op.sample_class != SC_GENERIC) {
        bool wide_enough = true; /* the body fetches both column taps with one two-texel load */
        for (int n = 0; n < op.nplanes; ++n)
            wide_enough = wide_enough && op.planes[n].w >= 2;
        if (wide_enough)
            op.fast_class = FC_SAMPLED;
    }
    if (op.fast_class == FC_NONE)
        return;
    /* the bodies of cghip_composite_fast that run this draw */
    const bool wide = op.planes[0].bits == 16;
    op.fetch_body   = op.fast_class == FC_SURFACE   ? (int)FB_SURFACE
                      : op.fast_class == FC_UYVY    ? (int)FB_UYVY8
                      : op.fast_class == FC_BYTES   ? (int)FB_BYTES8
                      : op.fast_class == FC_SAMPLED ? (op.sample_class == SC_UYVY8
                                           ? (int)FB_RECT_UYVY8 + ((op.flags & KF_UNIT_SCALE) ? 1 : 0)
                                           : (wide ? (int)FB_RECT16 : (int)FB_RECT8) +
                                                 (op.sample_class == SC_PACKED8 || op.sample_class == SC_PACKED16 ? 1 : 0) +
                                                 ((op.flags & KF_UNIT_SCALE) ? 2 : 0))
                      : op.fast_class == FC_PACKED  ? (wide ? (int)FB_PACKED16 : (int)FB_PACKED8)
                                                   : (wide ? (int)FB_PLANAR16 : (int)FB_PLANAR8) + op.csx + 2 * op.csy;
    const bool to_key = op.dst >= KS_KEY0;
    if (to_key && op.fast_class == FC_PLANAR && !(op.flags & (KF_CHROMA | KF_LEVELS | KF_CSB)))
        op.flags |= KF_RED_ONLY;
    const bool lk = op.local_key != KS_NONE, yk = op.layer_key != KS_NONE;
    const bool plain_keyer = !(op.flags & (KF_INVERT | KF_ADDITIVE));
    const bool simple = !lk && !yk && plain_keyer;
    op.blend_body = to_key                ? (simple ? (int)BB_KEY_SIMPLE : (int)BB_KEY_GENERAL)
                    : op.blend_mode != 0  ? (int)BB_MODE
                    : simple              ? (int)BB_RGBA_SIMPLE
                    : plain_keyer && lk != yk ? (int)BB_RGBA_ONE_KEY
                                          : (int)BB_RGBA_GENERAL;
}
static void classify_fast_draw(KOp& op, int frame_w)
{
    op.fast_class = FC_NONE;
    op.csx = op.csy = 0;
    op.fetch_body = op.blend_body = 0;
    op.flags &= ~(uint32_t)KF_RED_ONLY;
    if (op.kind == KOP_DRAW_SURFACE) {
        /* a temporary composited over the frame (image_mixer.cpp:207-225): no loads, so the only requirement is that
           lanes (4 pixels) are wholly inside or outside the box; pixels past the right frame edge are never stored */
        if ((op.x0 & 3) == 0 && ((op.x1 & 3) == 0 || op.x1 == frame_w))
            op.fast_class = FC_SURFACE;
        return;
    }
    /* item draws always carry blend mode 0 (the layer's mode is applied by its composite); checked anyway */
    if (op.kind != KOP_DRAW_PLANES || op.blend_mode != 0)
        return;
    const KPlane& p0 = op.planes[0];
    /* the colour producer's frame: one 1x1 bgra texel stretched over a rectangle (core/producer/color/
       color_producer.cpp). GL_LINEAR + CLAMP_TO_EDGE of a 1x1 texture returns that texel at every coordinate (both
       filters lerp a texel with itself), so the draw is a constant fill of its box whatever the scale */
    /* A lane owns 4 consecutive pixels. A box edge that is not on a 4-pixel column leaves a lane partly covered: it
       zeroes the source colour of its uncovered pixels (KF_EDGE), and a zero source leaves the target as it is - unless
       the draw inverts, which turns zero into one. */
    const bool edge = (op.x0 & 3) != 0 || ((op.x1 & 3) != 0 && op.x1 != frame_w);
    if (edge && (op.flags & KF_INVERT))
        return;
    if ((op.flags & KF_RECT) && op.nplanes == 1 && p0.w == 1 && p0.h == 1 && p0.channels == 4 &&
        op.pixel_format >= 1 && op.pixel_format <= 4 && (p0.bits == 8 || p0.bits == 16)) {
        op.flags |= KF_SOLID | (edge ? KF_EDGE : 0u);
        op.fast_class = FC_PACKED;
        return;
    }
    if (!(op.flags & KF_UNIT_SCALE))
        return;
    if ((p0.w & 3) != 0)
        return; /* rows must keep 4-texel groups aligned */
    /* every lane of the fast kernel owns 4 consecutive pixels and fetches them with one aligned vector load: the
       op's source offset must keep those groups whole, and the groups its box touches must be inside the plane */
    if ((op.unit_dx & 3) != 0)
        return;
    const int ax0 = op.x0 & ~3, ax1 = (op.x1 + 3) & ~3; /* what the lanes on the edges load */
    if (ax0 + op.unit_dx < 0 || ax1 + op.unit_dx > p0.w || op.y0 + op.unit_dy < 0 || op.y1 + op.unit_dy > p0.h)
        return;
    if (edge)
        op.flags |= KF_EDGE;
    const int pf = op.pixel_format;
    if ((pf >= 1 && pf <= 4) && op.nplanes == 1 && p0.channels == 4) {
        op.fast_class = FC_PACKED;
        return;
    }
    if ((pf == 0 || pf == 7 || pf == 8 || pf == 9) && op.nplanes == 1 && p0.bits == 8 &&
        p0.channels == ((pf == 8 || pf == 9) ? 3 : 1)) { /* gray, luma, bgr, rgb (shader.frag:485,499-505) */
        op.fast_class = FC_BYTES;
        return;
    }
    if (pf == 10 && op.nplanes == 2) { /* uyvy: both planes are views of one buffer (av_util.cpp:163-171) */
        const KPlane& p1 = op.planes[1];
        if (p0.bits == 8 && p1.bits == 8 && p0.channels == 2 && p1.channels == 4 && p1.w * 2 == p0.w && p1.h == p0.h &&
            p1.w >= 2 && (const char*)p1.ptr == (const char*)p0.ptr)
            op.fast_class = FC_UYVY;
        return;
    }
    if ((pf == 5 && op.nplanes == 3) || (pf == 6 && op.nplanes == 4)) {
        for (int n = 0; n < op.nplanes; ++n)
            if (op.planes[n].channels != 1 || op.planes[n].bits != p0.bits)
                return;
        const KPlane &cb = op.planes[1], &cr = op.planes[2];
        if (cb.w != cr.w || cb.h != cr.h)
            return;
        if (pf == 6 && (op.planes[3].w != p0.w || op.planes[3].h != p0.h))
            return;
        const bool fullx = cb.w == p0.w, halfx = cb.w * 2 == p0.w;
        const bool fully = cb.h == p0.h, halfy = cb.h * 2 == p0.h;
        if (!(fullx || halfx) || !(fully || halfy))
            return;
        if (halfx && (cb.w & 1) != 0)
            return;
        op.csx        = halfx ? 1 : 0;
        op.csy        = halfy ? 1 : 0;
        op.fast_class = FC_PLANAR;
        if (p0.bits == 16) { /* shader.frag:441-467 composed with code / 65535 and / 255, rounded once from double */
            const bool   hd = op.flags & KF_IS_HD;
            const double N = 65535.0, y0 = 1.164 * 16.0;
            const double krr = hd ? 1.793 : 1.596, kgr = hd ? 0.534 : 0.813, kgb = hd ? 0.213 : 0.391, kbb = hd ? 2.115 : 2.018;
            const double k[8] = {1.164 / N, krr / N, -(y0 + krr * 128.0) / 255.0, -kgr / N, -kgb / N,
                                 (-y0 + (kgr + kgb) * 128.0) / 255.0, kbb / N, -(y0 + kbb * 128.0) / 255.0};
            for (int i = 0; i < 8; ++i)
                op.ycoef[i] = (float)k[i];
        }
    }
}

/* which bilinear body of the general kernel fits the draw's source planes */
static int g_full_frame_composites = 0; /* test hook: composites of temporaries cover the frame, as in the reference */
struct pixel_box {
    int  x0 = INT_MAX, y0 = INT_MAX, x1 = INT_MIN, y1 = INT_MIN;
    bool empty() const { return x0 >= x1 || y0 >= y1; }
    void add(int ax0, int ay0, int ax1, int ay1)
    {
        if (ax0 >= ax1 || ay0 >= ay1)
            return;
        x0 = std::min(x0, ax0), y0 = std::min(y0, ay0), x1 = std::max(x1, ax1), y1 = std::max(y1, ay1);
    }
};
static int g_generic_sampling = 0; /* test hook: every draw takes the generic per-format sampling path */
static void classify_sampling(KOp& op)
{
    op.sample_class = SC_GENERIC;
    if (op.kind != KOP_DRAW_PLANES || g_generic_sampling)
        return;
    const int     pf = op.pixel_format;
    const KPlane& p0 = op.planes[0];
    if (p0.bits != 8 && p0.bits != 16)
        return;
    const bool wide = p0.bits == 16;
    if (!wide) /* the 8-bit bodies fetch both column taps of a row with one two-texel load */
        for (int n = 0; n < op.nplanes; ++n)
            if (op.planes[n].w < 2)
                return;
    if ((pf >= 1 && pf <= 4) && op.nplanes == 1 && p0.channels == 4) {
        op.sample_class = wide ? SC_PACKED16 : SC_PACKED8;
        return;
    }
    if (pf == 10 && op.nplanes == 2 && !wide && p0.channels == 2 && op.planes[1].channels == 4 && op.planes[1].bits == 8) {
        op.sample_class = SC_UYVY8; /* widths >= 2 checked above */
        return;
    }
    if ((pf == 5 && op.nplanes == 3) || (pf == 6 && op.nplanes == 4)) {
        for (int n = 0; n < op.nplanes; ++n)
            if (op.planes[n].channels != 1 || op.planes[n].bits != p0.bits)
                return;
        if (op.planes[1].w != op.planes[2].w || op.planes[1].h != op.planes[2].h)
            return;
        if (pf == 6 && (op.planes[3].w != p0.w || op.planes[3].h != p0.h))
            return;
        if (wide) /* 16-bit planar bodies fetch tap pairs too (packed 16-bit ones do not) */
            for (int n = 0; n < op.nplanes; ++n)
                if (op.planes[n].w < 2)
                    return;
        op.sample_class = wide ? SC_PLANAR16 : SC_PLANAR8;
    }
}

/* One reference draw -> one KOp. Returns 1 emitted, 0 skipped (reference early-out), <0 error. */
static int resolve_draw(const cghip_format& fmt, const cghip_draw& d, KOp& op)
{
    const double epsilon = 0.001; /* image_kernel.cpp:136 */
    const bool   from_surface = d.source_surface >= 0;
    if ((!from_surface && d.nplanes == 0) || d.target < 0) /* :140-142 */
        return 0;
    if (d.transform.opacity < epsilon) /* :144-146 */
        return 0;
    if (d.ncoords == 0) /* :150-152 */
        return 0;
    if (d.ncoords != 4)
        return fail(CGHIP_ERR_INVALID, "a quad needs exactly 4 coordinates (geometry.cpp:46-47)");
    vtx q[4];
    if (!transform_quad(d, q)) /* :220 */
        return 0;
    if (!d.geometry_is_quad) /* :378,413 */
        return 0;
    /* Every parallelogram gets q = 2 at all four corners (the diagonals bisect each other, :359-362), so the
       shader's TexCoord.st / TexCoord.q is an exact division by a constant: fold it into the map here and
       keep the per-pixel divide for genuinely projective quads only. */
    if ((float)q[0].q == (float)q[1].q && (float)q[1].q == (float)q[2].q && (float)q[2].q == (float)q[3].q &&
        (float)q[0].q != 0.0f && (float)q[0].q != 1.0f) {
        for (int i = 0; i < 4; ++i) {
            q[i].s = (double)((float)q[i].s / (float)q[i].q);
            q[i].t = (double)((float)q[i].t / (float)q[i].q);
            q[i].q = 1.0;
        }
    }

    const cghip_transform& t = d.transform;
    const int              W = fmt.width, H = fmt.height;
    std::memset(&op, 0, sizeof op);
    op.kind         = from_surface ? KOP_DRAW_SURFACE : KOP_DRAW_PLANES;
    op.pixel_format = from_surface ? (int)CGHIP_PF_BGRA : d.pixel_format;
    op.nplanes      = from_surface ? 0 : d.nplanes;
    op.blend_mode   = t.is_key ? CGHIP_BLEND_NORMAL : d.blend_mode; /* :270-272 */
    /* shader.frag:403,421: soft_light (12) and mix (>= 29) execute BlendNormal INSIDE `if (blend_mode != 0)`: the
       colour still goes through fore.rgb / (fore.a + 1e-7) * fore.a, which is not the identity (an inverted opaque
       source has a = 0, rgb != 0 and comes out black). So the mode is kept; get_blend_color's default is BlendNormal. */
    if (op.blend_mode < 0)
        op.blend_mode = 0;
    op.opacity = (float)(t.is_key ? 1.0 : t.opacity); /* :252 */
    if (d.keyer == CGHIP_KEYER_ADDITIVE)
        op.flags |= KF_ADDITIVE;
    if (t.invert)
        op.flags |= KF_INVERT;
    if (t.chroma_enable) { /* :254-266 */
        op.flags |= KF_CHROMA | (t.chroma_show_mask ? KF_CHROMA_MASK : 0u);
        op.ch_target_hue       = (float)(t.chroma_target_hue / 360.0);
        op.ch_hue_width        = (float)t.chroma_hue_width;
        op.ch_min_saturation   = (float)t.chroma_min_saturation;
        op.ch_min_brightness   = (float)t.chroma_min_brightness;
        op.ch_softness         = (float)(1.0 + t.chroma_softness);
        op.ch_spill            = (float)(t.chroma_spill_suppress / 360.0);
        op.ch_spill_saturation = (float)t.chroma_spill_suppress_saturation;
        op.ch_inv_spill        = 1.0f / op.ch_spill; /* +inf for 0: |diff| * inf keeps the shader's x/0 behaviour */
        op.ch_soft_range       = op.ch_softness - 1.0f;
        op.ch_inv_soft         = 1.0f / op.ch_soft_range;
    }
    if (t.levels_min_input > epsilon || t.levels_max_input < 1.0 - epsilon || t.levels_min_output > epsilon ||
        t.levels_max_output < 1.0 - epsilon || std::fabs(t.levels_gamma - 1.0) > epsilon) { /* :282-293 */
        op.flags |= KF_LEVELS;
        op.lv_min_input  = (float)t.levels_min_input;
        op.lv_range      = (float)t.levels_max_input - (float)t.levels_min_input;
        op.lv_inv_range  = 1.0f / op.lv_range;
        op.lv_inv_gamma  = 1.0f / (float)t.levels_gamma;
        op.lv_min_output = (float)t.levels_min_output;
        op.lv_max_output = (float)t.levels_max_output;
    }
    if (std::fabs(t.brightness - 1.0) > epsilon || std::fabs(t.saturation - 1.0) > epsilon ||
        std::fabs(t.contrast - 1.0) > epsilon) { /* :295-305 */
        op.flags |= KF_CSB;
        op.brt = (float)t.brightness;
        op.sat = (float)t.saturation;
        op.con = (float)t.contrast;
    }
    if (!from_surface) {
        for (int n = 0; n < d.nplanes; ++n) {
            const cghip_plane* p = d.planes[n];
            if (!p)
                return fail(CGHIP_ERR_INVALID, "draw references a null plane");
            op.planes[n] = KPlane{p->dptr, p->width, p->height, p->channels, p->bits, (float)p->width, (float)p->height};
        }
        if (d.planes[0]->height > 700) /* :248 */
            op.flags |= KF_IS_HD;
    } else if (H > 700) {
        op.flags |= KF_IS_HD; /* the temporary is frame-sized; unused by the bgra path anyway */
    }

    /* scissor :312-329 (window origin = memory row 0: FBO textures are not flipped) */
    int sx0 = 0, sy0 = 0, sx1 = W, sy1 = H;
    if (t.clip_translation[0] > DBL_EPSILON || t.clip_translation[1] > DBL_EPSILON ||
        t.clip_scale[0] < 1.0 - DBL_EPSILON || t.clip_scale[1] < 1.0 - DBL_EPSILON) {
        const int x = (int)(t.clip_translation[0] * (double)W), y = (int)(t.clip_translation[1] * (double)H);
        const int w = std::max(0, (int)(t.clip_scale[0] * (double)W)), h = std::max(0, (int)(t.clip_scale[1] * (double)H));
        sx0 = std::max(sx0, x), sy0 = std::max(sy0, y);
        sx1 = std::min(sx1, x + w), sy1 = std::min(sy1, y + h);
    }

    /* rasteriser setup */
    static const int idx[2][3] = {{0, 1, 2}, {0, 2, 3}}; /* :383-384 */
    raster_tri       tr[2]     = {setup_tri(q, idx[0], W, H), setup_tri(q, idx[1], W, H)};
    int64_t          xmin = INT64_MAX, xmax = INT64_MIN, ymin = INT64_MAX, ymax = INT64_MIN;
    op.ntri = 0;
    for (int k = 0; k < 2; ++k) {
        if (!tr[k].valid)
            continue;
        for (int i = 0; i < 3; ++i) {
            if (std::llabs(tr[k].X[i]) > (1ll << 24) || std::llabs(tr[k].Y[i]) > (1ll << 24))
                return fail(CGHIP_ERR_UNSUPPORTED, "quad vertex further than 65536 px from the frame origin");
            xmin = std::min(xmin, tr[k].X[i]), xmax = std::max(xmax, tr[k].X[i]);
            ymin = std::min(ymin, tr[k].Y[i]), ymax = std::max(ymax, tr[k].Y[i]);
        }
        emit_tri(tr[k], op.tri[op.ntri++]);
    }
    if (op.ntri == 0)
        return 0; /* zero-area quad: nothing rasterises */
    if (op.ntri == 1)
        op.tri[1] = op.tri[0];

    /* axis-aligned parallelogram with a single affine map: coverage is exactly a pixel rectangle */
    bool rect = tr[0].valid && tr[1].valid;
    if (rect) {
        const float vx[4] = {(float)q[0].vx, (float)q[1].vx, (float)q[2].vx, (float)q[3].vx};
        const float vy[4] = {(float)q[0].vy, (float)q[1].vy, (float)q[2].vy, (float)q[3].vy};
        const bool  a1 = vy[0] == vy[1] && vx[1] == vx[2] && vy[2] == vy[3] && vx[3] == vx[0];
        const bool  a2 = vx[0] == vx[1] && vy[1] == vy[2] && vx[2] == vx[3] && vy[3] == vy[0];
        rect           = a1 || a2;
        const double as[4][3] = {{(double)(float)q[0].s, (double)(float)q[0].t, (double)(float)q[0].q},
                                 {(double)(float)q[1].s, (double)(float)q[1].t, (double)(float)q[1].q},
                                 {(double)(float)q[2].s, (double)(float)q[2].t, (double)(float)q[2].q},
                                 {(double)(float)q[3].s, (double)(float)q[3].t, (double)(float)q[3].q}};
        for (int j = 0; j < 3 && rect; ++j)
            rect = std::fabs((as[0][j] + as[2][j]) - (as[1][j] + as[3][j])) < 1e-12;
    }
    bool projective = false;
    for (int i = 0; i < 4; ++i)
        projective = projective || (float)q[i].q != 1.0f;
    if (projective)
        op.flags |= KF_PROJECTIVE;
    /* s hangs on x alone and t on y alone: the fetch body for picture-in-picture boxes keeps t's side in scalars */
    if (rect && !projective && op.tri[0].a[0][2] == 0.0 && op.tri[0].a[1][1] == 0.0)
        op.flags |= KF_AXIS;

    int bx0, by0, bx1, by1;
    if (rect) {
        op.flags |= KF_RECT;
        bx0 = ceil_div256(xmin), bx1 = ceil_div256(xmax); /* left/top inclusive, right/bottom exclusive */
        by0 = ceil_div256(ymin), by1 = ceil_div256(ymax);
    } else {
        bx0 = floor_div256(xmin), bx1 = ceil_div256(xmax) + 1;
        by0 = floor_div256(ymin), by1 = ceil_div256(ymax) + 1;
    }
    op.x0 = std::max(bx0, sx0), op.x1 = std::min(bx1, sx1);
    op.y0 = std::max(by0, sy0), op.y1 = std::min(by1, sy1);
    if (op.x0 >= op.x1 || op.y0 >= op.y1)
        return 0; /* fully clipped: the GL draw touches no pixel */

    /* 1:1 mapping at an integer offset (full-screen video layers, crops, clips) */
    if (rect && !projective && !from_surface) {
        const KTri&  m  = op.tri[0];
        const double w0 = (double)op.planes[0].w, h0 = (double)op.planes[0].h;
        auto u = [&](double x, double y) { return (m.a[0][0] + m.a[0][1] * x + m.a[0][2] * y) * w0 - 0.5; };
        auto v = [&](double x, double y) { return (m.a[1][0] + m.a[1][1] * x + m.a[1][2] * y) * h0 - 0.5; };
        const double xa = op.x0, xb = op.x1 - 1, ya = op.y0, yb = op.y1 - 1;
        const double du = u(xa, ya) - xa, dv = v(xa, ya) - ya;
        /* How far from the texel grid a source may sit and still be addressed by texel index. 8-bit planes are
           filtered with 8-bit weights, rint(coordinate * 256) & 255: below 1/512 texel the weight is 0 and the result
           IS the texel, so 1e-3 is exact. 16-bit planes are filtered with fp32 weights (
           llvmpipe's float path): a weight w mixes in w * (neighbour - texel), up to w * 65535 LSB16, so the offset
           has to be below 2^-17 texel to stay under half an LSB16 on any content; anything further off takes the
           sampled path with the same fp32 weights as the reference pipeline. */
        bool wide = false;
        for (int n = 0; n < op.nplanes; ++n)
            wide = wide || op.planes[n].bits > 8;
        const double tol = wide ? 1.0 / 131072.0 : 1e-3;
        const bool   ok  = std::fabs(du - std::round(du)) < tol && std::fabs(dv - std::round(dv)) < tol &&
                        std::fabs((u(xb, yb) - xb) - du) < tol && std::fabs((v(xb, yb) - yb) - dv) < tol &&
                        std::fabs((u(xb, ya) - xb) - du) < tol && std::fabs((v(xa, yb) - yb) - dv) < tol &&
                        std::fabs((u(xa, yb) - xa) - du) < tol && std::fabs((v(xb, ya) - ya) - dv) < tol;
        if (ok) {
            op.flags |= KF_UNIT_SCALE;
            op.unit_dx = (int)std::llround(du);
            op.unit_dy = (int)std::llround(dv);
        }
    }
    return 1;
}

/* ---- draws nobody can see -----------------------------------------------------------------------------------
 * A full-frame layer whose source has no alpha channel (YCbCr, bgr / rgb, gray, luma), drawn at opacity 1 with the
 * linear keyer and no key, chroma key or invert, replaces the frame: fore.a = 1, so the keyer computes fore + 0 * back
 * (shader.frag:430-433), which does not depend on back. Everything drawn to the frame before it is invisible - a
 * looping background under a full-screen clip, say - and so is every key or temporary only such draws consumed. The
 * GL mixer draws it all; here a backward liveness pass over the op list drops it. */
static int g_keep_hidden_ops = 0; /* test hook */
static bool is_pseudo(const KOp& o);
static bool is_marker(const KOp& o);

static bool replaces_frame(const KOp& o, const cghip_format& fmt)
{
    const int pf = o.pixel_format;
    return o.kind == KOP_DRAW_PLANES && o.dst == KS_RGBA0 && (o.flags & KF_RECT) && o.x0 == 0 && o.y0 == 0 &&
           o.x1 == fmt.width && o.y1 == fmt.height && o.blend_mode == 0 && o.local_key == KS_NONE &&
           o.layer_key == KS_NONE && !(o.flags & (KF_INVERT | KF_ADDITIVE | KF_CHROMA)) && o.opacity == 1.0f &&
           (pf == 0 || pf == 5 || pf == 7 || pf == 8 || pf == 9 || pf == 10);
}

static void drop_hidden_ops(const cghip_format& fmt, frame_plan* out)
{
    if (g_keep_hidden_ops)
        return;
    std::vector<KOp>&    ops = out->ops;
    std::vector<uint8_t> keep(ops.size(), 0);
    uint32_t             live = 1u << KS_RGBA0; /* surfaces whose value at this point still matters */
    std::vector<int>     wanted;                /* parked key textures (surface slots) a kept KOP_LOAD_KEY brings back */
    bool                 any_dropped = false;
    for (size_t i = ops.size(); i-- > 0;) {
        const KOp&     o   = ops[i];
        const uint32_t dst = o.dst >= 0 && o.dst < KS_COUNT ? 1u << o.dst : 0u;
        if (o.kind == KOP_LOAD_KEY) { /* rewrites its register: what was there before is dead; needs its SAVE */
            if (!(live & dst)) {
                any_dropped = true;
                continue;
            }
            keep[i] = 1;
            live &= ~dst;
            wanted.push_back(o.src_surface);
            continue;
        }
        if (o.kind == KOP_SAVE_KEY) { /* kept iff the texture it parks comes back: then the register's value matters here */
            auto w = std::find(wanted.begin(), wanted.end(), o.src_surface);
            if (w == wanted.end()) {
                any_dropped = true;
                continue;
            }
            wanted.erase(w);
            keep[i] = 1;
            live |= dst;
            continue;
        }
        if (!(live & dst)) {
            any_dropped = true;
            continue;
        }
        keep[i] = 1;
        if (o.kind == KOP_CLEAR) {
            live &= ~dst;
            continue;
        }
        if (replaces_frame(o, fmt))
            live &= ~dst;
        for (int s : {o.local_key, o.layer_key, o.kind == KOP_DRAW_SURFACE ? o.src_surface : (int)KS_NONE})
            if (s >= 0 && s < KS_COUNT)
                live |= 1u << s;
    }
    if (!any_dropped)
        return;
    size_t n = 0;
    for (size_t i = 0; i < ops.size(); ++i)
        if (keep[i])
            ops[n++] = ops[i];
    ops.resize(n);
    out->full = false;
    out->fast = true;
    for (const KOp& o : ops) {
        if (o.dst == KS_RGBA1 || o.dst == KS_RGBA2)
            out->full = true;
        if (!is_marker(o) && o.fast_class == FC_NONE)
            out->fast = false;
    }
}

/* ---- segments: which kernel runs which run of ops --------------------------------------------------------------
 * The 4-pixels-per-lane kernel is about 3 times as fast per draw as the general one but takes only unit-scale
 * draws. A frame with one scaled box over a stack of full-screen layers should not pay the general price for the
 * whole stack, so the op list is cut into runs and every run gets its own launch; surfaces that are live across a
 * cut travel through HBM (KSpill). A cut costs a round trip of the frame (and a launch), which is worth about one
 * full-screen draw moved from the general kernel to the fast one: a two-state shortest path over the op list decides
 * where cutting pays. */
static int g_force_general = 0;
static int g_keep_pseudo_ops = 0; /* test hook: KOP_SAVE_KEY / KOP_LOAD_KEY stay in the op list that cghip_debug_plan returns */
static int g_max_segments  = 8;
static double g_cut_scale  = 1.0;

static uint32_t surf_bit(int s) { return s >= 0 && s < KS_COUNT ? 1u << s : 0u; }
static bool     is_pseudo(const KOp& o) { return o.kind == KOP_SAVE_KEY || o.kind == KOP_LOAD_KEY; }
static bool     is_marker(const KOp& o) { return o.kind == KOP_CLEAR || is_pseudo(o); } /* no pixels of its own */
static uint32_t refs_of(const KOp& o) /* surfaces an op reads or writes */
{
    if (is_marker(o)) /* CLEAR / LOAD_KEY rewrite their register, SAVE_KEY reads it */
        return surf_bit(o.dst);
    return surf_bit(o.dst) | surf_bit(o.local_key) | surf_bit(o.layer_key) |
           (o.kind == KOP_DRAW_SURFACE ? surf_bit(o.src_surface) : 0u) |
           (o.fused ? surf_bit(o.b_local_key) | surf_bit(o.b_layer_key) : 0u);
}

static void split_into_segments(const cghip_format& fmt, frame_plan* out)
{
    const std::vector<KOp>& ops = out->ops;
    const int               n   = (int)ops.size();
    const int               W = fmt.width, H = fmt.height;
    out->segments.clear();
    out->spill_mask = 0;

    /* 1. kernel class per op: min-cost path, states 0 = general, 1 = fast. Costs in picoseconds, measured with
       tests/gpu_segments_probe.py and tests/gpu_probe.py: per pixel of a draw's box a layer
       takes about 4.8 ps in the general kernel and 1.6 (16-bit) / 1.9 (8-bit) ps in the fast one; a cut costs a round
       trip of the frame through HBM (3.1 / 1.6 ps per frame pixel) plus the gap between two launches (see `gap` below). */
    std::vector<uint8_t> cls((size_t)n, out->fast && !g_force_general ? 1 : 0);
    if (n > 0 && !out->fast && !g_force_general && g_max_segments > 1) {
        /* tests/gpu_probe.py, 16 layers 2160p 8-bit, ps per pixel of the draw's box. General kernel: axis-aligned 4.6 -
           4.9, perspective 5.4, rotated 6.3, composite of a temporary with a blend mode 4.4. Fast kernel: aligned 1:1
           1.9 (16-bit 1.6), sampled box of planes 3.5 (5.0 into a temporary, where the kernel holds 128 registers),
           of a packed plane 2.6 (16-bit planes: 3.9 and 3.2), composite of a temporary 3.3. */
        /* gap: what a cut costs beyond the bytes. Two launches back to back are 6 us apart, but the general kernel's time
           is not proportional to its draws' pixels (every wave pays set-up, surface loads and stores): BASELINE config 2
           cut in two - the first three layers to the fast kernel, predicted gain 7 us - ran 0.102 ms against 0.086 ms in
           one general launch. 15 us keeps such marginal cuts away and every cut the tests and probes show to pay. */
        const double rt_px = fmt.bits == 16 ? 3.1 : 1.6, gap = 15e6;
        auto per_px = [&](const KOp& o, int k) -> double {
            const bool planes = o.kind == KOP_DRAW_PLANES, temp = o.dst == KS_RGBA1 || o.dst == KS_RGBA2;
            const double fused = o.fused ? (k == 0 ? 1.5 : 1.7) : 0.0; /* the composite that rides on the item */
            if (k == 0)
                return (o.kind == KOP_DRAW_SURFACE ? 4.4 : (planes && !(o.flags & KF_RECT)) ? 6.0 : 4.8) + fused;
            if (planes && o.fetch_body >= FB_RECT8 && o.fetch_body < FB_SAMPLED_END)
                return (o.fetch_body >= FB_RECT_UYVY8 ? 3.2 : ((o.fetch_body - FB_RECT8) & 1 ? 2.6 : 3.5) * (o.fetch_body >= FB_RECT16 ? 1.15 : 1.0)) +
                       (temp ? 1.5 : 0.0) + fused;
            if (o.kind == KOP_DRAW_SURFACE)
                return o.blend_mode != 0 ? 3.3 : 1.9;
            return (fmt.bits == 16 ? 1.6 : 1.9) + fused;
        };
        /* into the fast kernel the whole frame makes the trip; into the general one, which may cover only the box of
           its draws, this draw's box stands in for that */
        auto cut_into = [&](int k, double area) { return (gap + rt_px * (k == 1 ? (double)W * H : area)) * g_cut_scale; };
        const double inf       = 1e300;
        std::vector<double>  cost((size_t)(n + 1) * 2, inf);
        std::vector<uint8_t> from((size_t)(n + 1) * 2, 0);
        cost[0] = cost[1] = 0.0;
        auto box_area = [&](int i) { /* a CLEAR in front of a draw is priced as that draw when it opens a run */
            while (i < n && is_marker(ops[(size_t)i]))
                ++i;
            return i < n ? (double)std::max(0, ops[(size_t)i].x1 - ops[(size_t)i].x0) *
                               std::max(0, ops[(size_t)i].y1 - ops[(size_t)i].y0)
                         : 0.0;
        };
        for (int i = 0; i < n; ++i) {
            const KOp&   o    = ops[(size_t)i];
            const bool   clr  = is_marker(o);
            const double area = clr ? 0.0 : box_area(i), open_area = box_area(i);
            for (int k = 0; k < 2; ++k) {
                if (k == 1 && !clr && o.fast_class == FC_NONE)
                    continue;
                for (int j = 0; j < 2; ++j) {
                    const double c = cost[(size_t)i * 2 + j] + (j != k ? cut_into(k, open_area) : 0.0) + area * per_px(o, k);
                    if (c < cost[(size_t)(i + 1) * 2 + k]) {
                        cost[(size_t)(i + 1) * 2 + k] = c;
                        from[(size_t)(i + 1) * 2 + k] = (uint8_t)j;
                    }
                }
            }
        }
        int k = cost[(size_t)n * 2 + 1] < cost[(size_t)n * 2] ? 1 : 0;
        for (int i = n; i > 0; --i) {
            cls[(size_t)i - 1] = (uint8_t)k;
            k                  = from[(size_t)i * 2 + k];
        }
        int cuts = 0;
        for (int i = 1; i < n; ++i)
            cuts += cls[(size_t)i] != cls[(size_t)i - 1];
        if (cuts + 1 > g_max_segments)
            std::fill(cls.begin(), cls.end(), (uint8_t)0);
    }

    for (int i = n - 2; i >= 0; --i) /* a marker runs with the draw it stands in front of */
        if (is_marker(ops[(size_t)i]))
            cls[(size_t)i] = cls[(size_t)i + 1];
    /* 2. runs of one class. A key texture that leaves or re-enters a register (KOP_SAVE_KEY / KOP_LOAD_KEY: more than
       two key textures alive at once) forces a cut in front of the group of markers it stands in: the launch before the
       cut stores the register, the launch behind it loads the register. */
    std::vector<uint8_t> forced((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i)
        if (is_pseudo(ops[(size_t)i])) {
            int g = i;
            while (g > 0 && is_marker(ops[(size_t)g - 1]))
                --g;
            forced[(size_t)g] = 1;
        }
    for (int i = 0; i < n;) {
        segment sg;
        sg.first = i;
        sg.fast  = cls[(size_t)i] != 0;
        do
            ++i;
        while (i < n && (cls[(size_t)i] != 0) == sg.fast && !forced[(size_t)i]);
        sg.count = i - sg.first;
        out->segments.push_back(sg);
    }
    if (out->segments.empty()) {
        segment sg;
        sg.fast = !g_force_general; /* empty scene: the zero frame (image_mixer.cpp:89-92) */
        out->segments.push_back(sg);
    }

    /* 3. which surfaces cross the cuts. in_mem: the HBM copy is the current value; a surface whose last write in a
       segment is a CLEAR is zero, and zero is what a launch's registers start from, so it needs no copy. The RGBA
       temporaries have a buffer each; KEY0 / KEY1 take theirs from a small set (key_bufs): a key texture parked in HBM
       (KOP_SAVE_KEY) keeps the buffer it was stored in while its register serves another texture. */
    const uint32_t KEYS = surf_bit(KS_KEY0) | surf_bit(KS_KEY1);
    const size_t nseg   = out->segments.size();
    uint32_t     in_mem = 0;
    int          key_buf_of[2] = {-1, -1};         /* buffer holding the register's current value, if in_mem says so */
    std::vector<int>    home;                      /* surface slot -> buffer of its parked key texture (-2: it is zero) */
    std::vector<uint8_t> buf_used;                 /* buffer ids in use */
    auto buf_alloc = [&]() {
        for (size_t b = 0; b < buf_used.size(); ++b)
            if (!buf_used[b]) {
                buf_used[b] = 1;
                return (int)b;
            }
        buf_used.push_back(1);
        return (int)buf_used.size() - 1;
    };
    auto buf_free = [&](int b) {
        if (b >= 0)
            buf_used[(size_t)b] = 0;
    };
    auto home_of = [&](int slot) -> int& {
        if ((size_t)slot >= home.size())
            home.resize((size_t)slot + 1, -1);
        return home[(size_t)slot];
    };
    for (size_t si = 0; si < nseg; ++si) {
        segment& sg      = out->segments[si];
        uint32_t touched = 0, written = 0, zero = 0, later = surf_bit(KS_RGBA0);
        /* key textures that re-enter a register at the head of this launch */
        for (int i = sg.first; i < sg.first + sg.count && is_marker(ops[(size_t)i]); ++i) {
            const KOp& o = ops[(size_t)i];
            if (o.kind != KOP_LOAD_KEY)
                continue;
            const int r = o.dst - KS_KEY0, b = home_of(o.src_surface);
            if (in_mem & surf_bit(o.dst)) /* what the register held is dead (it was saved or is not needed) */
                buf_free(key_buf_of[r]);
            in_mem &= ~surf_bit(o.dst);
            key_buf_of[r] = -1;
            if (b >= 0) {
                in_mem |= surf_bit(o.dst);
                key_buf_of[r] = b;
            }
            home_of(o.src_surface) = -1;
        }
        for (int i = sg.first; i < sg.first + sg.count; ++i) {
            const KOp& o = ops[(size_t)i];
            if (is_pseudo(o)) { /* SAVE belongs to the launch before, LOAD was dealt with: the register counts as read */
                if (o.kind == KOP_LOAD_KEY)
                    touched |= surf_bit(o.dst);
                continue;
            }
            touched |= refs_of(o);
            written |= surf_bit(o.dst);
            if (o.kind == KOP_CLEAR)
                zero |= surf_bit(o.dst);
            else
                zero &= ~surf_bit(o.dst);
        }
        uint32_t reset = 0; /* surfaces a later launch clears before anything looks at them: dead at this cut */
        for (int i = sg.first + sg.count; i < n; ++i) {
            const KOp& o = ops[(size_t)i];
            if (o.kind == KOP_CLEAR || o.kind == KOP_LOAD_KEY)
                reset |= surf_bit(o.dst) & ~later;
            else
                later |= refs_of(o) & ~reset; /* a SAVE_KEY reads its register */
        }
        sg.full       = (touched & (surf_bit(KS_RGBA1) | surf_bit(KS_RGBA2))) != 0;
        sg.ext        = 0;
        sg.fused      = false;
        for (int i = sg.first; i < sg.first + sg.count; ++i)
            sg.fused = sg.fused || ops[(size_t)i].fused != 0;
        for (int i = sg.first; i < sg.first + sg.count && sg.fast; ++i) {
            const KOp& o = ops[(size_t)i];
            if (o.kind == KOP_DRAW_PLANES)
                sg.ext = std::max(sg.ext, o.fused ? 2 : (o.fetch_body >= FB_RECT8 && o.fetch_body < FB_SAMPLED_END) ? 1 : 0);
            if (o.kind == KOP_DRAW_SURFACE && o.fused) /* the composite of a MIX pair that is all its blend-mode layer holds */
                sg.ext = 2;
        }
        sg.mix_pairs = false;
        for (int i = sg.first; i < sg.first + sg.count; ++i)
            sg.mix_pairs = sg.mix_pairs || (ops[(size_t)i].kind == KOP_DRAW_PLANES && ops[(size_t)i].blend_body == BB_MIX_HEAD);
        sg.yuv_lut = false;
        for (int i = sg.first; i < sg.first + sg.count && sg.fast; ++i) {
            const KOp& o = ops[(size_t)i];
            sg.yuv_lut = sg.yuv_lut || (o.kind == KOP_DRAW_PLANES && ((o.fetch_body >= FB_PLANAR8 && o.fetch_body < FB_PLANAR8 + 4) || o.fetch_body == FB_UYVY8 ||
                                                                      o.fetch_body == FB_RECT8 || o.fetch_body == FB_RECT8 + 2 || /* sampled boxes of planes */
                                                                      o.fetch_body == FB_RECT_UYVY8 || o.fetch_body == FB_RECT_UYVY8 + 1));
        }
        sg.mono = -1; /* one source format throughout: the kernel instance compiled for that fetch body alone */
        if (sg.fast && sg.ext == 0) { /* launches with temporaries: the format of the draws that fetch from planes */
            int body = -2;
            for (int i = sg.first; i < sg.first + sg.count; ++i) {
                const KOp& o = ops[(size_t)i];
                if (is_marker(o) || (sg.full && o.kind == KOP_DRAW_SURFACE))
                    continue;
                body = (body == -2 || body == o.fetch_body) ? o.fetch_body : -1;
            }
            sg.mono = body >= 0 ? body : -1;
        }
        sg.load_mask  = touched & in_mem;
        sg.store_mask = written & ~zero & later;
        if (si == 0)
            sg.store_mask |= surf_bit(KS_RGBA0); /* the first launch defines every pixel of the frame */
        sg.one_temp = sg.full && !((touched | sg.load_mask | sg.store_mask) & surf_bit(KS_RGBA2)) ? 1 : 0;
        if (sg.one_temp) {
            bool modes = false;
            for (int i = sg.first; i < sg.first + sg.count; ++i) {
                const KOp& o = ops[(size_t)i];
                modes = modes || (!is_marker(o) && (o.blend_mode != 0 || o.fused || o.blend_body == BB_MODE || o.blend_body == BB_FUSED));
            }
            if (!modes)
                sg.one_temp = 2;
        }
        for (int r = 0; r < 2; ++r) { /* the key registers' buffers */
            const uint32_t bit = surf_bit(KS_KEY0 + r);
            sg.key_ld[r] = (sg.load_mask & bit) ? key_buf_of[r] : -1;
            if (written & bit) { /* the register's memory copy, if any, is stale */
                buf_free(key_buf_of[r]);
                key_buf_of[r] = -1;
            }
            if (sg.store_mask & bit)
                sg.key_st[r] = key_buf_of[r] = buf_alloc();
        }
        in_mem = (in_mem & ~written) | sg.store_mask;
        out->spill_mask |= (sg.load_mask | sg.store_mask) & ~surf_bit(KS_RGBA0) & ~KEYS;
        /* key textures that leave their register behind this launch: they keep the buffer the value is in */
        if (si + 1 < nseg) {
            const segment& nx = out->segments[si + 1];
            for (int i = nx.first; i < nx.first + nx.count && is_marker(ops[(size_t)i]); ++i) {
                const KOp& o = ops[(size_t)i];
                if (o.kind != KOP_SAVE_KEY)
                    continue;
                const int r = o.dst - KS_KEY0;
                if (in_mem & surf_bit(o.dst)) {
                    home_of(o.src_surface) = key_buf_of[r]; /* the buffer changes owner */
                } else {
                    home_of(o.src_surface) = -2; /* cleared and never drawn into: zero, which is what a load-less register is */
                }
                in_mem &= ~surf_bit(o.dst);
                key_buf_of[r] = -1;
            }
        }

        /* 4. a general launch that hands nothing but the frame itself onwards only has to cover its draws */
        sg.gx0 = sg.gy0 = 0, sg.gx1 = W, sg.gy1 = H;
        if (si > 0 && !sg.fast && (sg.store_mask & ~surf_bit(KS_RGBA0)) == 0) {
            int bx0 = W, by0 = H, bx1 = 0, by1 = 0;
            for (int i = sg.first; i < sg.first + sg.count; ++i) {
                const KOp& o = ops[(size_t)i];
                if (is_marker(o))
                    continue;
                bx0 = std::min(bx0, o.x0), by0 = std::min(by0, o.y0);
                bx1 = std::max(bx1, o.x1), by1 = std::max(by1, o.y1);
            }
            bx0 = std::max(bx0, 0) & ~63, by0 = std::max(by0, 0) & ~3; /* whole waves, whole workgroups */
            bx1 = std::min(bx1, W), by1 = std::min(by1, H);
            if (bx0 < bx1 && by0 < by1)
                sg.gx0 = bx0, sg.gy0 = by0, sg.gx1 = bx1, sg.gy1 = by1;
        }
    }
    out->n_key_bufs = (int)buf_used.size();

    /* 5. the markers have done their work: the kernels never see them */
    if (!g_keep_pseudo_ops) {
        std::vector<KOp>& mops = out->ops;
        std::vector<int>  new_index((size_t)n + 1, 0);
        int               m = 0;
        for (int i = 0; i < n; ++i) {
            new_index[(size_t)i] = m;
            if (!is_pseudo(mops[(size_t)i]))
                mops[(size_t)m++] = mops[(size_t)i];
        }
        new_index[(size_t)n] = m;
        mops.resize((size_t)m);
        for (segment& sg : out->segments) {
            const int a = new_index[(size_t)sg.first], b = new_index[(size_t)(sg.first + sg.count)];
            sg.first = a, sg.count = b - a;
        }
    }
}

/* ---- blend-mode layers with one item, in launches of the general kernel ----------------------------------------
 * CLEAR T; item -> T; composite T -> D with the layer's mode: three ops, two of them walking a full register surface
 * through per-lane selects, and a frame that needs the kernel variant with the RGBA temporaries. Fused (KOp::fused) the
 * item's colour is blended over zero and rounded to the UNORM code the texture would have held - the same arithmetic
 * with a literal zero background -, then goes through the composite's blend over D. Conditions: the item is the only
 * draw into T since its CLEAR and sits right in front of the composite; nothing reads T afterwards before its next
 * CLEAR; the composite has no colour adjustments (it never has: its transform is the default one). Done before the
 * frame is cut into launches: a fused op is one op for either kernel, and a frame whose blend-mode layers all hold one
 * item no longer needs the kernel variants with the RGBA temporaries. Exact: tests compare fused and unfused frames bit
 * for bit (test hook cghip_debug_no_layer_fusion). BASELINE config 2: 20 ops -> 8. */
static int g_no_layer_fusion = 0;

static void fuse_layers(const cghip_format& fmt, frame_plan* out)
{
    if (g_no_layer_fusion)
        return;
    std::vector<KOp>& ops = out->ops;
    const int         n   = (int)ops.size();
    std::vector<char> drop((size_t)n, 0);
    bool              any = false;
    for (int j = 1; j < n; ++j) {
        const KOp& B = ops[(size_t)j];
        const KOp& A = ops[(size_t)j - 1];
        if (B.kind != KOP_DRAW_SURFACE || A.kind != KOP_DRAW_PLANES || drop[(size_t)j - 1])
            continue;
        const int T = B.src_surface;
        if ((T != KS_RGBA1 && T != KS_RGBA2) || A.dst != T || B.dst == T || A.fused || A.blend_mode != 0)
            continue;
        if (B.flags & (KF_LEVELS | KF_CSB | KF_CHROMA))
            continue;
        /* the item is the only draw into T since T's CLEAR */
        int c = j - 2;
        for (; c >= 0; --c) {
            const KOp& o = ops[(size_t)c];
            if (drop[(size_t)c])
                continue;
            if (o.kind == KOP_CLEAR && o.dst == T)
                break;
            if (refs_of(o) & surf_bit(T))
                c = -1;
        }
        if (c < 0)
            continue;
        /* nothing looks at T after the composite before T is cleared again */
        bool dead = true;
        for (int k = j + 1; k < n && dead; ++k) {
            const KOp& o = ops[(size_t)k];
            if (o.kind == KOP_CLEAR && o.dst == T)
                break;
            dead = !(refs_of(o) & surf_bit(T));
        }
        if (!dead)
            continue;
        /* The 4-pixels-per-lane kernels run a fused op (blend body BB_FUSED) when the lanes of the composite's box are
           the lanes that touch the item's box - the composite was shrunk to the item's box widened to 4-pixel columns
           (plan_frame; not for the HSL modes, whose composite covers the frame) - and the mode is a real one. An item
           of their class that does not meet this stays three ops: fused it would belong to the general kernel. */
        const bool same_lanes = B.x0 == (A.x0 & ~3) && B.x1 == std::min(fmt.width, (A.x1 + 3) & ~3) && B.y0 == A.y0 && B.y1 == A.y1;
        const bool fast_fused = A.fast_class != FC_NONE && same_lanes && B.blend_mode != 0 && B.dst >= KS_RGBA0 && B.dst <= KS_RGBA2;
        if (A.fast_class != FC_NONE && !fast_fused)
            continue;
        KOp F = A;
        F.fused = 1;
        F.ax0 = A.x0, F.ay0 = A.y0, F.ax1 = A.x1, F.ay1 = A.y1;
        F.x0 = B.x0, F.y0 = B.y0, F.x1 = B.x1, F.y1 = B.y1;
        F.dst = B.dst;
        F.b_blend_mode = B.blend_mode, F.b_local_key = B.local_key, F.b_layer_key = B.layer_key;
        F.b_flags = B.flags, F.b_opacity = B.opacity;
        if (fast_fused)
            F.blend_body = BB_FUSED;
        else
            F.fast_class = FC_NONE;
        ops[(size_t)j]      = F;
        drop[(size_t)j - 1] = 1;
        drop[(size_t)c]     = 1; /* T's CLEAR: nothing between it and the composite touched T but the item */
        any                 = true;
    }
    if (!any)
        return;
    size_t m = 0;
    for (int i = 0; i < n; ++i)
        if (!drop[(size_t)i])
            ops[m++] = ops[(size_t)i];
    ops.resize(m);
    out->full = false;
    out->fast = true;
    for (const KOp& o : ops) {
        if (refs_of(o) & (surf_bit(KS_RGBA1) | surf_bit(KS_RGBA2)))
            out->full = true;
        if (!is_marker(o) && o.fast_class == FC_NONE)
            out->fast = false;
    }
}

/* A MIX transition outside blend-mode layers - two is_mix items drawn into the cleared mix texture with the additive
 * keyer, the texture composited over the frame (image_mixer.cpp:187-225; BASELINE config 4, every crossfade of a
 * rundown) - is four ops: CLEAR T, A -> T, B -> T, T -> frame. With T in registers the frame needs the kernel variants
 * that hold a temporary (96 - 128 registers, 4 - 5 waves per SIMD). Fused, the three draws are ONE step of the lean
 * kernel: A over zero, its UNORM store, B added, its UNORM store, the result composited - the same arithmetic in the
 * same order (test hook cghip_debug_no_mix_fusion; tests/test_fuzz.py compares the two plans bit for bit) with the
 * texture in registers of that step alone. Only for frames that are the 4-pixels-per-lane class throughout and need no
 * temporary once their pairs are fused; both items aligned unit-scale draws of one box, the composite plain. */
static int g_no_mix_fusion = 0;
static void fuse_mix_pairs(const cghip_format& fmt, frame_plan* out)
{
    if (g_no_mix_fusion || g_force_general || !out->fast || !out->full)
        return;
    const std::vector<KOp>& ops = out->ops;
    const int               n   = (int)ops.size();
    bool has_ext = false; /* sampled boxes or fused single-item layers elsewhere in the frame: the step then exists in the
                             kernel without temporaries only (cghip_composite_fast_xp has every body), see below */
    for (const KOp& o : ops)
        has_ext = has_ext || (!is_marker(o) && (o.fused || (o.kind == KOP_DRAW_PLANES && o.fetch_body >= FB_RECT8 && o.fetch_body < FB_SAMPLED_END)));
    const uint32_t          TEMPS = surf_bit(KS_RGBA1) | surf_bit(KS_RGBA2);
    std::vector<int>        head;   /* index of the CLEAR of every pair that qualifies */
    std::vector<char>       in_pair((size_t)n, 0), has_clear((size_t)n, 0);
    for (int c = 0; c + 3 < n; ++c) {
        const KOp &C = ops[(size_t)c], &A = ops[(size_t)c + 1], &B = ops[(size_t)c + 2];
        if (C.kind != KOP_CLEAR || (C.dst != KS_RGBA1 && C.dst != KS_RGBA2))
            continue;
        const int T = C.dst;
        auto item = [&](const KOp& o) {
            return o.kind == KOP_DRAW_PLANES && o.dst == T && o.blend_mode == 0 && !o.fused &&
                   (o.fast_class == FC_PACKED || o.fast_class == FC_PLANAR || o.fast_class == FC_UYVY || o.fast_class == FC_BYTES) &&
                   !(refs_of(o) & ~(surf_bit(T) | surf_bit(KS_KEY0) | surf_bit(KS_KEY1)));
        };
        if (!item(A) || !item(B) || A.x0 != B.x0 || A.y0 != B.y0 || A.x1 != B.x1 || A.y1 != B.y1)
            continue;
        /* inside a blend-mode layer the layer texture is cleared between the items and the composite (the resolver
           creates it when the first thing is drawn into it, image_mixer.cpp:118-129): that CLEAR stays where it is */
        const bool c2 = ops[(size_t)c + 3].kind == KOP_CLEAR && c + 4 < n;
        const KOp& S  = ops[(size_t)c + 3 + (c2 ? 1 : 0)];
        if (c2 && (ops[(size_t)c + 3].dst == T || ops[(size_t)c + 3].dst != S.dst))
            continue;
        if (S.kind != KOP_DRAW_SURFACE || S.src_surface != T || S.dst == T || S.dst > KS_RGBA2 || S.blend_mode != 0 || S.fused ||
            S.fast_class != FC_SURFACE || (S.flags & (KF_LEVELS | KF_CSB | KF_CHROMA | KF_INVERT)))
            continue;
        if (S.x0 != (A.x0 & ~3) || S.x1 != std::min(fmt.width, (A.x1 + 3) & ~3) || S.y0 != A.y0 || S.y1 != A.y1)
            continue; /* the composite's lanes are the items' lanes */
        const int last = c + 3 + (c2 ? 1 : 0);
        bool      dead = true; /* nothing looks at T afterwards before it is cleared again */
        for (int k = last + 1; k < n && dead; ++k) {
            const KOp& o = ops[(size_t)k];
            if (o.kind == KOP_CLEAR && o.dst == T)
                break;
            dead = !(refs_of(o) & surf_bit(T));
        }
        if (!dead)
            continue;
        head.push_back(c);
        has_clear[(size_t)c] = c2;
        in_pair[(size_t)c] = in_pair[(size_t)c + 1] = in_pair[(size_t)c + 2] = in_pair[(size_t)last] = 1;
        c = last;
    }
    if (head.empty())
        return;
    /* A pair that is all its blend-mode layer holds - CLEAR L; the pair -> L; L -> frame with the layer's mode, L dead
       afterwards - can hand its result to the mode's blend in registers as well (KOp::fused on the pair's composite,
       which takes the layer composite's b_* fields like a fused single item, fuse_layers): over the cleared L the
       composite's result is its fore colour through the UNORM store. Then the frame needs no temporary at all. Taken
       only if that holds for every pair composited into a temporary and nothing else uses one: the blend call exists in
       the step of the kernel without temporaries (cghip_composite_fast_xp), not of the one that holds the layer. */
    std::vector<char> layer_fused((size_t)n, 0), gone((size_t)n, 0);
    bool              all_layers = !g_no_layer_fusion;
    for (int c : head) {
        const int  s_at = c + 3 + (has_clear[(size_t)c] ? 1 : 0);
        const KOp& S    = ops[(size_t)s_at];
        if (S.dst == KS_RGBA0)
            continue;
        bool ok = has_clear[(size_t)c] && s_at + 1 < n;
        if (ok) {
            const KOp& M = ops[(size_t)s_at + 1];
            ok = M.kind == KOP_DRAW_SURFACE && M.src_surface == S.dst && M.dst == KS_RGBA0 && M.blend_mode != 0 && !M.fused &&
                 M.fast_class == FC_SURFACE && !(M.flags & (KF_LEVELS | KF_CSB | KF_CHROMA)) && M.x0 == S.x0 && M.x1 == S.x1 &&
                 M.y0 == S.y0 && M.y1 == S.y1;
            for (int k = s_at + 2; k < n && ok; ++k) { /* nothing looks at L afterwards before it is cleared again */
                const KOp& o = ops[(size_t)k];
                if (o.kind == KOP_CLEAR && o.dst == S.dst)
                    break;
                ok = !(refs_of(o) & surf_bit(S.dst));
            }
        }
        if (!ok) {
            all_layers = false;
            break;
        }
        layer_fused[(size_t)c] = 1;
        gone[(size_t)c + 3] = gone[(size_t)s_at + 1] = 1; /* L's CLEAR and the layer's composite */
    }
    if (all_layers)
        for (int i = 0; i < n && all_layers; ++i)
            if (!in_pair[(size_t)i] && !gone[(size_t)i] && (refs_of(ops[(size_t)i]) & TEMPS))
                all_layers = false;
    if (!all_layers) {
        std::fill(layer_fused.begin(), layer_fused.end(), 0);
        std::fill(gone.begin(), gone.end(), 0);
    }
    /* All or nothing, and at most ONE temporary left: the step exists in the kernel without temporaries and in the one
       that holds a single temporary (KSpill::one_temp), not in the one with two at its 128-register bound. What is left
       are the targets of the pairs' composites and whatever the other ops use. */
    uint32_t left = 0;
    for (int i = 0; i < n; ++i)
        if (!in_pair[(size_t)i] && !gone[(size_t)i])
            left |= refs_of(ops[(size_t)i]) & TEMPS;
    for (int c : head)
        if (!layer_fused[(size_t)c])
            left |= surf_bit(ops[(size_t)c + 3 + (has_clear[(size_t)c] ? 1 : 0)].dst) & TEMPS;
    if (left == TEMPS || (left != 0 && has_ext))
        return;
    auto rename = [&](int32_t& surf) { /* the one temporary left is RGBA1 to the kernel */
        if (left == surf_bit(KS_RGBA2) && surf == KS_RGBA2)
            surf = KS_RGBA1;
    };
    std::vector<KOp> fused;
    fused.reserve((size_t)n);
    size_t h = 0;
    for (int i = 0; i < n; ++i) {
        if (h < head.size() && i == head[h]) {
            const bool c2 = has_clear[(size_t)i] != 0;
            KOp A = ops[(size_t)i + 1], B = ops[(size_t)i + 2], S = ops[(size_t)i + 3 + (c2 ? 1 : 0)];
            if (layer_fused[(size_t)i]) {
                const KOp& M = ops[(size_t)i + 5];
                S.fused = 1;
                S.ax0 = S.x0, S.ay0 = S.y0, S.ax1 = S.x1, S.ay1 = S.y1;
                S.dst = M.dst;
                S.b_blend_mode = M.blend_mode, S.b_local_key = M.local_key, S.b_layer_key = M.layer_key;
                S.b_flags = M.flags, S.b_opacity = M.opacity;
                ++i; /* M */
            } else if (c2) {
                KOp C2 = ops[(size_t)i + 3];
                rename(C2.dst);
                fused.push_back(C2);
            }
            rename(S.dst);
            A.blend_body = BB_MIX_HEAD;
            A.dst = B.dst = S.dst;   /* what the step as a whole writes */
            S.src_surface = S.dst;   /* never read as a source: the step hands the texture over in registers */
            fused.push_back(A), fused.push_back(B), fused.push_back(S);
            i += 3 + (c2 ? 1 : 0);
            ++h;
            continue;
        }
        KOp o = ops[(size_t)i];
        if (!is_pseudo(o)) {
            rename(o.dst);
            if (o.kind == KOP_DRAW_SURFACE)
                rename(o.src_surface);
        }
        fused.push_back(o);
    }
    out->ops.swap(fused);
    out->full = left != 0;
    out->mix_pairs = true;
}

/* Per launch: which of its ops can touch which band of the frame. The kernels used to walk the whole op list in
 * every wave and cull by box there; a wave of a 1080p frame with eight scaled boxes paid the scalar set-up of every op
 * (~100 scalar instructions each, on a scalar unit shared by four SIMDs) to find that six of them are elsewhere.
 * Column bands are one wave wide (64 pixels; 256 in the 4-pixels-per-lane kernel), row bands one workgroup high (4 rows).
 * CLEARs are in every band (they reset a register surface wherever it is used). */
static void op_masks(const cghip_format& fmt, frame_plan* out)
{
    out->masks.clear();
    for (segment& sg : out->segments) {
        sg.col_shift = sg.fast ? 8 : 6;
        sg.ncols     = ((fmt.width - 1) >> sg.col_shift) + 1;
        sg.nrows     = ((fmt.height - 1) >> 2) + 1;
        sg.nchunks   = std::max(1, (sg.count + 63) / 64);
        sg.mask_off  = out->masks.size();
        const size_t stride = (size_t)sg.ncols + (size_t)sg.nrows;
        out->masks.resize(out->masks.size() + stride * (size_t)sg.nchunks, 0);
        uint64_t* base = out->masks.data() + sg.mask_off;
        for (int i = 0; i < sg.count; ++i) {
            const KOp&     o   = out->ops[(size_t)(sg.first + i)];
            uint64_t*      m   = base + stride * (size_t)(i / 64);
            const uint64_t bit = 1ull << (i % 64);
            int            c0 = 0, c1 = sg.ncols - 1, r0 = 0, r1 = sg.nrows - 1;
            if (o.kind != KOP_CLEAR) {
                if (o.x0 >= o.x1 || o.y0 >= o.y1)
                    continue;
                c0 = std::max(0, o.x0) >> sg.col_shift, c1 = std::min(sg.ncols - 1, (o.x1 - 1) >> sg.col_shift);
                r0 = std::max(0, o.y0) >> 2, r1 = std::min(sg.nrows - 1, (o.y1 - 1) >> 2);
            }
            for (int c = c0; c <= c1; ++c)
                m[c] |= bit;
            for (int r = r0; r <= r1; ++r)
                m[sg.ncols + r] |= bit;
        }
        /* every op in every band (full-screen stacks): the kernel then does not read the table at all */
        sg.dense = true;
        for (int k = 0; k < sg.nchunks && sg.dense; ++k) {
            const int      nbits = std::min(64, sg.count - 64 * k);
            const uint64_t all   = nbits >= 64 ? ~0ull : ((1ull << std::max(0, nbits)) - 1);
            for (size_t j = 0; j < stride; ++j)
                if (base[stride * (size_t)k + j] != all) {
                    sg.dense = false;
                    break;
                }
        }
    }
}

int plan_frame(const cghip_format* fmt, const cghip_surface* surfaces, int nsurfaces, const cghip_draw* draws,
               int ndraws, frame_plan* out)
{
    out->ops.clear();
    out->planes.clear();
    out->full = false;
    out->fast = false;
    if (!fmt || fmt->width <= 0 || fmt->height <= 0 || (fmt->bits != 8 && fmt->bits != 16))
        return fail(CGHIP_ERR_INVALID, "bad cghip_format");
    if (nsurfaces < 1 || nsurfaces > CGHIP_MAX_SURFACES || !surfaces || surfaces[0].channels != 4)
        return fail(CGHIP_ERR_INVALID, "surface 0 must be the 4-channel frame target");
    if (ndraws > 0 && !draws)
        return fail(CGHIP_ERR_INVALID, "null draw list");

    /* pass 1: resolve every draw */
    std::vector<KOp> resolved((size_t)ndraws);
    std::vector<int> emitted((size_t)ndraws, 0);
    auto             slot_ok = [&](int s, bool allow_none) { return (allow_none && s < 0) || (s >= 0 && s < nsurfaces); };
    for (int i = 0; i < ndraws; ++i) {
        const cghip_draw& d = draws[i];
        if (!slot_ok(d.target, false) || !slot_ok(d.local_key, true) || !slot_ok(d.layer_key, true) ||
            !slot_ok(d.source_surface, true))
            return fail(CGHIP_ERR_INVALID, "draw references a surface slot out of range");
        if ((d.local_key >= 0 && surfaces[d.local_key].channels != 1) ||
            (d.layer_key >= 0 && surfaces[d.layer_key].channels != 1) ||
            (d.source_surface >= 0 && surfaces[d.source_surface].channels != 4))
            return fail(CGHIP_ERR_INVALID, "key surfaces have 1 channel, source surfaces 4");
        if (d.source_surface < 0 && (d.nplanes < 0 || d.nplanes > 4))
            return fail(CGHIP_ERR_INVALID, "nplanes out of range");
        const int rc = resolve_draw(*fmt, d, resolved[(size_t)i]);
        if (rc < 0)
            return rc;
        emitted[(size_t)i] = rc;
    }

    /* pass 2: last use of every surface slot among emitted draws */
    std::vector<int> last_use((size_t)nsurfaces, -1);
    for (int i = 0; i < ndraws; ++i) {
        if (!emitted[(size_t)i])
            continue;
        const cghip_draw& d = draws[i];
        for (int s : {d.target, d.local_key, d.layer_key, d.source_surface})
            if (s >= 0)
                last_use[(size_t)s] = i;
    }

    /* pass 3: bind slots to register surfaces, emitting a CLEAR at first touch (fresh textures are cleared,
       device.cpp:207-209) */
    std::vector<int> reg((size_t)nsurfaces, KS_NONE);
    int              owner[KS_COUNT];
    pixel_box        drawn[KS_COUNT]; /* what has been drawn into a register surface since its CLEAR */
    for (int& o : owner)
        o = -1;
    reg[0]          = KS_RGBA0;
    owner[KS_RGBA0] = 0;
    std::vector<uint8_t> parked((size_t)nsurfaces, 0); /* key textures that sit in HBM (KOP_SAVE_KEY) */
    auto marker = [&](int kind, int r, int slot) {
        KOp c;
        std::memset(&c, 0, sizeof c);
        c.kind = kind, c.dst = r, c.src_surface = slot;
        c.local_key = c.layer_key = c.b_local_key = c.b_layer_key = KS_NONE;
        out->ops.push_back(c);
    };
    auto bind = [&](int slot, int at) -> int {
        if (slot < 0)
            return KS_NONE;
        if (reg[(size_t)slot] != KS_NONE)
            return reg[(size_t)slot];
        const bool key = surfaces[slot].channels != 4;
        const int  lo = key ? KS_KEY0 : KS_RGBA1, hi = key ? KS_KEY1 : KS_RGBA2;
        int        r = -1;
        for (int k = lo; k <= hi && r < 0; ++k) {
            if (owner[k] >= 0 && last_use[(size_t)owner[k]] < at) { /* previous tenant is dead */
                reg[(size_t)owner[k]] = KS_NONE;
                owner[k]              = -1;
            }
            if (owner[k] < 0)
                r = k;
        }
        if (r < 0 && key) {
            /* Both key registers hold textures that are still needed: more than two key textures are alive at once
               (a parent's pending layer key + a sublayer's layer key + its local key, image_mixer.cpp:104-114,153). The
               one this draw does not use and that is needed again last goes to HBM; it comes back when a draw names it. */
            const cghip_draw& d = draws[at];
            int               best = -1, best_next = -1;
            for (int k = lo; k <= hi; ++k) {
                const int tenant = owner[k];
                if (tenant == d.target || tenant == d.local_key || tenant == d.layer_key || tenant == d.source_surface)
                    continue;
                int next = ndraws;
                for (int j = at; j < ndraws; ++j)
                    if (emitted[(size_t)j] && (draws[j].target == tenant || draws[j].local_key == tenant ||
                                               draws[j].layer_key == tenant || draws[j].source_surface == tenant)) {
                        next = j;
                        break;
                    }
                if (next > best_next)
                    best = k, best_next = next;
            }
            if (best >= 0) {
                marker(KOP_SAVE_KEY, best, owner[best]);
                parked[(size_t)owner[best]] = 1;
                reg[(size_t)owner[best]]    = KS_NONE;
                owner[best]                 = -1;
                r                           = best;
            }
        }
        if (r < 0)
            return -100;
        owner[r]          = slot;
        reg[(size_t)slot] = r;
        if (key && parked[(size_t)slot]) { /* back from HBM: no CLEAR, the launch loads it */
            parked[(size_t)slot] = 0;
            marker(KOP_LOAD_KEY, r, slot);
            return r;
        }
        marker(KOP_CLEAR, r, KS_NONE);
        drawn[r] = pixel_box();
        if (r == KS_RGBA1 || r == KS_RGBA2)
            out->full = true;
        return r;
    };
    for (int i = 0; i < ndraws; ++i) {
        if (!emitted[(size_t)i])
            continue;
        const cghip_draw& d  = draws[i];
        KOp&              op = resolved[(size_t)i];
        /* inputs first: a key consumed by this draw must not lose its register to this draw's target */
        op.src_surface = bind(d.source_surface, i);
        op.local_key   = bind(d.local_key, i);
        op.layer_key   = bind(d.layer_key, i);
        op.dst         = bind(d.target, i);
        if (op.src_surface == -100 || op.local_key == -100 || op.layer_key == -100 || op.dst == -100)
            return fail(CGHIP_ERR_UNSUPPORTED, /* the reference's renderer never has more than a layer and a mix texture alive
                                                    (image_mixer.cpp:128,187); key textures are unbounded and handled above */
                        "frame needs more than 2 live RGBA temporaries at once");
        if (surfaces[d.target].channels == 1 && d.source_surface >= 0)
            return fail(CGHIP_ERR_INVALID, "surface-to-key draws do not exist in the reference renderer");
        /* A temporary is composited with a full-frame draw (image_mixer.cpp:207-225), but outside of what was drawn
           into it since its CLEAR it is zero, and a zero source leaves the target as it is: fore = 0, the blend
           colour is multiplied by fore.a = 0, and back passes through both keyers unchanged (shader.frag:424-434).
           So the draw only has to cover that box (widened to whole 4-pixel groups for the 4-pixels-per-lane kernel).
           Not for the HSL modes, whose blend colour can be inf for unpremultiplied backgrounds, nor with invert or
           colour adjustments, which turn a zero source into something else. */
        if (op.kind == KOP_DRAW_SURFACE && !(op.blend_mode >= 25 && op.blend_mode <= 28) &&
            !(op.flags & (KF_INVERT | KF_LEVELS | KF_CSB | KF_CHROMA)) && !g_full_frame_composites) {
            const pixel_box& d = drawn[op.src_surface];
            if (d.empty())
                continue; /* nothing was drawn into the temporary */
            op.x0 = std::max(op.x0, d.x0 & ~3), op.x1 = std::min(op.x1, std::min(fmt->width, (d.x1 + 3) & ~3));
            op.y0 = std::max(op.y0, d.y0), op.y1 = std::min(op.y1, d.y1);
            if (op.x0 >= op.x1 || op.y0 >= op.y1)
                continue;
        }
        drawn[op.dst].add(op.x0, op.y0, op.x1, op.y1);
        op.ax0 = op.x0, op.ay0 = op.y0, op.ax1 = op.x1, op.ay1 = op.y1;
        op.b_local_key = op.b_layer_key = KS_NONE;
        classify_sampling(op);
        classify_fast(op, fmt->width); /* dst is known now */
        out->ops.push_back(op);
        if (d.source_surface < 0)
            for (int n = 0; n < d.nplanes; ++n)
                out->planes.push_back(const_cast<cghip_plane*>(d.planes[n]));
    }
    out->fast = true; /* with out->full: the variant of the fast kernel that also holds the two RGBA temporaries */
    for (const KOp& o : out->ops)
        if (!is_marker(o) && o.fast_class == FC_NONE)
            out->fast = false;
    drop_hidden_ops(*fmt, out);
    fuse_layers(*fmt, out);
    fuse_mix_pairs(*fmt, out);
    split_into_segments(*fmt, out);
    op_masks(*fmt, out);
    return CGHIP_OK;
}

} // namespace cghip

/* Introspection for host-logic tests (no GPU needed): plans a frame and copies out the KOps. */
extern "C" int cghip_debug_plan(const cghip_format* fmt, const cghip_surface* surfaces, int nsurfaces,
                                const cghip_draw* draws, int ndraws, void* kops_out, int capacity, int* nops,
                                int* full)
{
    cghip::frame_plan plan;
    const int         rc = cghip::plan_frame(fmt, surfaces, nsurfaces, draws, ndraws, &plan);
    if (rc != CGHIP_OK)
        return rc;
    if (nops)
        *nops = (int)plan.ops.size();
    if (full)
        *full = (plan.full ? 1 : 0) | (plan.fast ? 2 : 0) | ((int)plan.segments.size() << 8);
    if (kops_out) {
        const int n = std::min(capacity, (int)plan.ops.size());
        if (n > 0) /* an empty vector's data() may be null, which memcpy must not be handed even for 0 bytes */
            std::memcpy(kops_out, plan.ops.data(), (size_t)n * sizeof(KOp));
    }
    return CGHIP_OK;
}

extern "C" int cghip_debug_kop_size(void) { return (int)sizeof(KOp); }

/* test hooks: one launch of the general kernel per frame / at most n launches per frame (1 = never cut) */
extern "C" void cghip_debug_force_general_kernel(int on) { cghip::g_force_general = on; }
/* KOP_SAVE_KEY / KOP_LOAD_KEY markers stay in the op list cghip_debug_plan returns (never while rendering) */
extern "C" void cghip_debug_keep_pseudo_ops(int on) { cghip::g_keep_pseudo_ops = on; }
extern "C" void cghip_debug_generic_sampling(int on) { cghip::g_generic_sampling = on; }
/* sampled (scaled / rotated / perspective) draws stay with the 1-pixel-per-lane kernel, as before FC_SAMPLED existed */
extern "C" void cghip_debug_no_sampled_fast(int on) { cghip::g_no_sampled_fast = on; }
/* blend-mode layers with one item stay three ops (CLEAR, item, composite) in launches of the general kernel */
extern "C" void cghip_debug_no_layer_fusion(int on) { cghip::g_no_layer_fusion = on; }
/* MIX pairs stay four ops (CLEAR, two items, composite) in the kernels that hold a temporary */
extern "C" void cghip_debug_no_mix_fusion(int on) { cghip::g_no_mix_fusion = on; }
extern "C" void cghip_debug_keep_hidden_ops(int on) { cghip::g_keep_hidden_ops = on; }
extern "C" void cghip_debug_full_frame_composites(int on) { cghip::g_full_frame_composites = on; }
extern "C" void cghip_debug_max_segments(int n) { cghip::g_max_segments = n >= 1 ? n : 8; }
/* scales what the planner thinks a cut costs: 0 cuts at every change of class (tests: small frames never cut otherwise) */
extern "C" void cghip_debug_cut_cost_scale(double scale) { cghip::g_cut_scale = scale >= 0.0 ? scale : 1.0; }

/* Introspection: the launches of a planned frame, 14 ints each: first, count, fast, full, load_mask, store_mask,
 * gx0, gy0, gx1, gy1, then the key buffers KEY0 / KEY1 are loaded from and stored to (-1: none). Returns the number of
 * segments. */
extern "C" int cghip_debug_plan_segments(const cghip_format* fmt, const cghip_surface* surfaces, int nsurfaces,
                                         const cghip_draw* draws, int ndraws, int* out, int capacity)
{
    cghip::frame_plan plan;
    const int         rc = cghip::plan_frame(fmt, surfaces, nsurfaces, draws, ndraws, &plan);
    if (rc != CGHIP_OK)
        return rc;
    int k = 0;
    for (const cghip::segment& sg : plan.segments) {
        if (k >= capacity)
            break;
        const int v[14] = {sg.first, sg.count, sg.fast, sg.full, (int)sg.load_mask, (int)sg.store_mask, sg.gx0, sg.gy0, sg.gx1, sg.gy1,
                           sg.key_ld[0], sg.key_ld[1], sg.key_st[0], sg.key_st[1]};
        std::memcpy(out + (size_t)k * 14, v, sizeof v);
        ++k;
    }
    return (int)plan.segments.size();
}
