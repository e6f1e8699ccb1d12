/*
 * oracle/ref_scene.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Plain-C restatement of the reference image mixer's HOST logic, backend-agnostic:
 *   core/frame/frame_transform.cpp:32-108          image_transform::operator*=
 *   accelerator/ogl/image/image_mixer.cpp:54-72    item / layer
 *   accelerator/ogl/image/image_mixer.cpp:86-226   image_renderer (layer walk, key/mix/blend temporaries)
 *   accelerator/ogl/image/image_mixer.cpp:247-304  push / visit / pop / render
 *   accelerator/ogl/image/image_kernel.cpp:46-107,134-375   per-draw coordinate + uniform math
 * It emits ref_draw records to a ref_backend (llvmpipe or the CPU restatement).
 *
 * Parity status: the reference has no tests, so this host restatement is pinned only
 * through the pixels it produces: goldens in tests/golden/ come from THIS file driving the reference
 * shaders under llvmpipe. The host math itself is "parity unpinned" by any reference fixture.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "ref_types.h"

/* ------------------------------------------------------------------ transforms */

void ref_transform_default(ref_transform* t)
{ /* frame_transform.h:34-99 defaults */
    memset(t, 0, sizeof *t);
    t->opacity = t->contrast = t->brightness = t->saturation = 1.0;
    t->fill_scale[0] = t->fill_scale[1] = 1.0;
    t->clip_scale[0] = t->clip_scale[1] = 1.0;
    t->crop_lr[0] = t->crop_lr[1] = 1.0;
    t->pers_ur[0] = 1.0;
    t->pers_lr[0] = t->pers_lr[1] = 1.0;
    t->pers_ll[1] = 1.0;
    t->lv_max_input = t->lv_gamma = t->lv_max_output = 1.0;
    t->chroma_spill_suppress_saturation = 1.0;
}

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* self *= other  (frame_transform.cpp:53-108) */
void ref_transform_compose(ref_transform* s, const ref_transform* o)
{
    s->opacity *= o->opacity;
    s->brightness *= o->brightness;
    s->contrast *= o->contrast;
    s->saturation *= o->saturation;

    const double aspect_ratio = 1.0; /* :61 */
    double       orig_x = o->fill_translation[0];
    double       orig_y = o->fill_translation[1] / aspect_ratio;
    double       rot0   = orig_x * cos(s->angle) - orig_y * sin(s->angle);
    double       rot1   = orig_x * sin(s->angle) + orig_y * cos(s->angle);
    rot1 *= aspect_ratio;

    s->anchor[0] += o->anchor[0] * s->fill_scale[0];
    s->anchor[1] += o->anchor[1] * s->fill_scale[1];
    s->fill_translation[0] += rot0 * s->fill_scale[0];
    s->fill_translation[1] += rot1 * s->fill_scale[1];
    s->fill_scale[0] *= o->fill_scale[0];
    s->fill_scale[1] *= o->fill_scale[1];
    s->clip_translation[0] += o->clip_translation[0] * s->clip_scale[0];
    s->clip_translation[1] += o->clip_translation[1] * s->clip_scale[1];
    s->clip_scale[0] *= o->clip_scale[0];
    s->clip_scale[1] *= o->clip_scale[1];
    s->angle += o->angle;

    /* transform_rect(crop) :33-39 */
    s->crop_ul[0] += o->crop_ul[0];
    s->crop_ul[1] += o->crop_ul[1];
    s->crop_lr[0] *= o->crop_lr[0];
    s->crop_lr[1] *= o->crop_lr[1];
    /* transform_corners(perspective) :41-51 */
    s->pers_ul[0] += o->pers_ul[0];
    s->pers_ul[1] += o->pers_ul[1];
    s->pers_lr[0] *= o->pers_lr[0];
    s->pers_lr[1] *= o->pers_lr[1];
    s->pers_ur[0] *= o->pers_ur[0];
    s->pers_ur[1] += o->pers_ur[1];
    s->pers_ll[0] += o->pers_ll[0];
    s->pers_ll[1] *= o->pers_ll[1];

    s->lv_min_input  = dmax(s->lv_min_input, o->lv_min_input);
    s->lv_max_input  = dmin(s->lv_max_input, o->lv_max_input);
    s->lv_min_output = dmax(s->lv_min_output, o->lv_min_output);
    s->lv_max_output = dmin(s->lv_max_output, o->lv_max_output);
    s->lv_gamma *= o->lv_gamma;
    s->chroma_enable |= o->chroma_enable;
    s->chroma_show_mask |= o->chroma_show_mask;
    s->chroma_target_hue     = dmax(o->chroma_target_hue, s->chroma_target_hue);
    s->chroma_min_saturation = dmax(o->chroma_min_saturation, s->chroma_min_saturation);
    s->chroma_min_brightness = dmax(o->chroma_min_brightness, s->chroma_min_brightness);
    s->chroma_hue_width      = dmax(o->chroma_hue_width, s->chroma_hue_width);
    s->chroma_softness       = dmax(o->chroma_softness, s->chroma_softness);
    s->chroma_spill_suppress = dmax(o->chroma_spill_suppress, s->chroma_spill_suppress);
    s->chroma_spill_suppress_saturation =
        dmin(o->chroma_spill_suppress_saturation, s->chroma_spill_suppress_saturation);
    s->is_key |= o->is_key;
    s->invert |= o->invert;
    s->is_mix |= o->is_mix;
    s->blend_mode = s->blend_mode > o->blend_mode ? s->blend_mode : o->blend_mode;
    s->layer_depth += o->layer_depth;
}

void ref_geometry_default(ref_coord c[4])
{ /* geometry.cpp:64-76 */
    static const double v[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    for (int i = 0; i < 4; ++i) {
        c[i].vertex_x = c[i].texture_x = v[i][0];
        c[i].vertex_y = c[i].texture_y = v[i][1];
        c[i].texture_r                 = 0.0;
        c[i].texture_q                 = 1.0;
    }
}

void ref_geometry_default_vflip(ref_coord c[4])
{ /* geometry.cpp:77-89 */
    ref_geometry_default(c);
    c[0].texture_y = c[1].texture_y = 1.0;
    c[2].texture_y = c[3].texture_y = 0.0;
}

static int coords_equal(const ref_coord* a, const ref_coord* b)
{ /* geometry.cpp:35-39, over 4 coords */
    for (int i = 0; i < 4; ++i)
        if (!(a[i].vertex_x == b[i].vertex_x && a[i].vertex_y == b[i].vertex_y && a[i].texture_x == b[i].texture_x &&
              a[i].texture_y == b[i].texture_y && a[i].texture_r == b[i].texture_r &&
              a[i].texture_q == b[i].texture_q))
            return 0;
    return 1;
}

/* ------------------------------------------------------------------ scene graph */

typedef struct ref_item {
    int32_t       format, nplanes;
    void*         textures[4];
    ref_transform transform;
    int32_t       is_quad, ncoords;
    ref_coord     coords[4];
} ref_item;

typedef struct ref_layer {
    struct ref_layer* sublayers;
    int               nsub, capsub;
    ref_item*         items;
    int               nitems, capitems;
    int               blend_mode;
} ref_layer;

typedef struct ref_mixer {
    const ref_backend* be;
    ref_transform*     tstack;
    int                ntstack, captstack;
    ref_layer*         layers; /* top-level */
    int                nlayers, caplayers;
    /* layer_stack_ holds raw pointers into vectors in the reference (:236); vectors may move here too,
       so keep index paths instead: path[d] = index at depth d. */
    int  path[64];
    int  npath;
    void** owned; /* textures created at visit; destroyed after render */
    int    nowned, capowned;
    int    draws_issued;
    /* optional draw log for host-logic tests */
    ref_draw* log;
    int       nlog, caplog;
    int       log_enabled;
} ref_mixer;

static void* grow(void* p, int* cap, int need, size_t elem)
{
    if (need <= *cap)
        return p;
    int nc = *cap ? *cap * 2 : 8;
    while (nc < need)
        nc *= 2;
    p    = realloc(p, (size_t)nc * elem);
    *cap = nc;
    return p;
}

ref_mixer* ref_mixer_create(const ref_backend* be)
{
    ref_mixer* m = (ref_mixer*)calloc(1, sizeof *m);
    m->be        = be;
    m->tstack    = (ref_transform*)grow(NULL, &m->captstack, 1, sizeof(ref_transform));
    ref_transform_default(&m->tstack[0]); /* transform_stack_(1) :242 */
    m->ntstack = 1;
    return m;
}

static void layer_free(ref_layer* l)
{
    for (int i = 0; i < l->nsub; ++i)
        layer_free(&l->sublayers[i]);
    free(l->sublayers);
    free(l->items);
}

static void scene_clear(ref_mixer* m)
{
    for (int i = 0; i < m->nlayers; ++i)
        layer_free(&m->layers[i]);
    m->nlayers = 0;
    for (int i = 0; i < m->nowned; ++i)
        m->be->tex_destroy(m->owned[i]);
    m->nowned = 0;
}

void ref_mixer_destroy(ref_mixer* m)
{
    scene_clear(m);
    free(m->layers);
    free(m->tstack);
    free(m->owned);
    free(m->log);
    free(m);
}

static ref_layer* current_layer(ref_mixer* m)
{
    if (m->npath == 0)
        return NULL;
    ref_layer* l = &m->layers[m->path[0]];
    for (int d = 1; d < m->npath; ++d)
        l = &l->sublayers[m->path[d]];
    return l;
}

void ref_mixer_push(ref_mixer* m, const ref_transform* image_transform)
{ /* image_mixer.cpp:247-264 */
    int prev_depth = m->tstack[m->ntstack - 1].layer_depth;
    m->tstack      = (ref_transform*)grow(m->tstack, &m->captstack, m->ntstack + 1, sizeof(ref_transform));
    m->tstack[m->ntstack] = m->tstack[m->ntstack - 1];
    ref_transform_compose(&m->tstack[m->ntstack], image_transform);
    m->ntstack++;
    int new_depth = m->tstack[m->ntstack - 1].layer_depth;

    if (prev_depth < new_depth) {
        ref_layer nl;
        memset(&nl, 0, sizeof nl);
        nl.blend_mode = m->tstack[m->ntstack - 1].blend_mode;
        if (m->npath == 0) {
            m->layers = (ref_layer*)grow(m->layers, &m->caplayers, m->nlayers + 1, sizeof(ref_layer));
            m->layers[m->nlayers] = nl;
            m->path[m->npath++]   = m->nlayers++;
        } else if (m->npath < 64) {
            ref_layer* cur = current_layer(m);
            cur->sublayers = (ref_layer*)grow(cur->sublayers, &cur->capsub, cur->nsub + 1, sizeof(ref_layer));
            cur->sublayers[cur->nsub] = nl;
            m->path[m->npath++]       = cur->nsub++;
        }
    }
}

/* returns 0 ok / 1 skipped / -1 error (no open layer: the reference would dereference an empty stack) */
int ref_mixer_visit(ref_mixer* m, const ref_frame* f)
{ /* image_mixer.cpp:266-293 */
    if (f->format == REF_PF_INVALID)
        return 1;
    if (f->nplanes == 0)
        return 1;
    ref_layer* cur = current_layer(m);
    if (!cur)
        return -1;
    ref_item it;
    memset(&it, 0, sizeof it);
    it.format    = f->format;
    it.nplanes   = f->nplanes;
    it.transform = m->tstack[m->ntstack - 1];
    it.is_quad   = f->geometry_is_quad;
    it.ncoords   = f->ncoords;
    memcpy(it.coords, f->coords, sizeof it.coords);
    for (int n = 0; n < f->nplanes; ++n) { /* device.cpp:242-262 copy_async(array -> texture) */
        const ref_plane* p = &f->planes[n];
        void*            t = m->be->tex_create(p->width, p->height, p->stride, p->bits);
        m->be->tex_upload(t, p->data);
        it.textures[n] = t;
        m->owned       = (void**)grow(m->owned, &m->capowned, m->nowned + 1, sizeof(void*));
        m->owned[m->nowned++] = t;
    }
    cur->items = (ref_item*)grow(cur->items, &cur->capitems, cur->nitems + 1, sizeof(ref_item));
    cur->items[cur->nitems++] = it;
    return 0;
}

void ref_mixer_pop(ref_mixer* m)
{ /* image_mixer.cpp:295-299 */
    if (m->ntstack > 1)
        m->ntstack--;
    int depth = m->tstack[m->ntstack - 1].layer_depth;
    if (depth < 0)
        depth = 0;
    if (depth < m->npath)
        m->npath = depth; /* layer_stack_.resize(depth): only ever shrinks in practice */
}

/* ------------------------------------------------------------------ image_kernel::draw host part */

static int get_line_intersection(double p0_x, double p0_y, double p1_x, double p1_y, double p2_x, double p2_y,
                                 double p3_x, double p3_y, double* rx, double* ry)
{ /* image_kernel.cpp:46-74 */
    double s1_x = p1_x - p0_x, s1_y = p1_y - p0_y;
    double s2_x = p3_x - p2_x, s2_y = p3_y - p2_y;
    double s    = (-s1_y * (p0_x - p2_x) + s1_x * (p0_y - p2_y)) / (-s2_x * s1_y + s1_x * s2_y);
    double t    = (s2_x * (p0_y - p2_y) - s2_y * (p0_x - p2_x)) / (-s2_x * s1_y + s1_x * s2_y);
    if (s >= 0 && s <= 1 && t >= 0 && t <= 1) {
        *rx = p0_x + t * s1_x;
        *ry = p0_y + t * s1_y;
        return 1;
    }
    return 0;
}

static double hypotenuse(double x1, double y1, double x2, double y2)
{ /* :76-82 */
    double x = x2 - x1, y = y2 - y1;
    return sqrt(x * x + y * y);
}

static double calc_q(double close_diagonal, double distant_diagonal)
{ /* :84-87 */
    return (close_diagonal + distant_diagonal) / distant_diagonal;
}

typedef struct kernel_params { /* image_kernel.h:40-52 */
    int32_t              format, nplanes;
    void*                textures[4];
    const ref_transform* transform;
    int                  is_quad, ncoords;
    const ref_coord*     coords;
    int                  blend_mode, keyer;
    void *               background, *local_key, *layer_key;
    double               aspect_ratio;
    int                  field_mode;
} kernel_params;

/* Returns 1 if a draw was emitted into *out, 0 if the reference would early-out. */
static int kernel_resolve(const ref_backend* be, const kernel_params* p, ref_draw* out)
{
    const double         epsilon = 0.001; /* :136 */
    const ref_transform* t       = p->transform;

    if (p->nplanes == 0 || !p->background) /* :140-142 */
        return 0;
    if (t->opacity < epsilon) /* :144-146 */
        return 0;
    if (p->ncoords == 0) /* :150-152 */
        return 0;

    ref_coord coords[4];
    memcpy(coords, p->coords, sizeof coords);

    ref_coord def[4], defv[4];
    ref_geometry_default(def);
    ref_geometry_default_vflip(defv);
    int is_default_geometry = coords_equal(coords, def) || coords_equal(coords, defv); /* :158-159 */

    double aspect = p->aspect_ratio, angle = t->angle;
    double pers[4][2] = {{t->pers_ul[0], t->pers_ul[1]},
                         {t->pers_ur[0] - 1.0, t->pers_ur[1]},
                         {t->pers_lr[0] - 1.0, t->pers_lr[1] - 1.0},
                         {t->pers_ll[0], t->pers_ll[1] - 1.0}}; /* :164-169 */

    for (int i = 0; i < 4; ++i) { /* :207-217 */
        ref_coord* c = &coords[i];
        if (is_default_geometry) { /* do_crop :171-185 */
            c->vertex_x  = dmin(dmax(c->vertex_x, t->crop_ul[0]), t->crop_lr[0]);
            c->vertex_y  = dmin(dmax(c->vertex_y, t->crop_ul[1]), t->crop_lr[1]);
            c->texture_x = dmin(dmax(c->texture_x, t->crop_ul[0]), t->crop_lr[0]);
            c->texture_y = dmin(dmax(c->texture_y, t->crop_ul[1]), t->crop_lr[1]);
            /* do_perspective :186-194 */
            c->vertex_x += pers[i][0];
            c->vertex_y += pers[i][1];
        }
        /* rotate :195-201 */
        double ox   = (c->vertex_x - t->anchor[0]) * t->fill_scale[0];
        double oy   = (c->vertex_y - t->anchor[1]) * t->fill_scale[1] / aspect;
        c->vertex_x = ox * cos(angle) - oy * sin(angle);
        c->vertex_y = ox * sin(angle) + oy * cos(angle);
        c->vertex_y *= aspect;
        /* move :202-205 */
        c->vertex_x += t->fill_translation[0];
        c->vertex_y += t->fill_translation[1];
    }

    { /* is_outside_screen :97-107,220 */
        int l = 1, r = 1, a = 1, b = 1;
        for (int i = 0; i < 4; ++i) {
            l &= coords[i].vertex_x < 0.0;
            r &= coords[i].vertex_x > 1.0;
            a &= coords[i].vertex_y < 0.0;
            b &= coords[i].vertex_y > 1.0;
        }
        if (l || r || a || b)
            return 0;
    }

    memset(out, 0, sizeof *out);
    out->pixel_format = p->format;
    out->nplanes      = p->nplanes;
    for (int n = 0; n < p->nplanes; ++n)
        out->planes[n] = p->textures[n];
    out->background    = p->background;
    out->local_key     = p->local_key;
    out->layer_key     = p->layer_key;
    out->is_hd         = be->tex_height(p->textures[0]) > 700 ? 1 : 0; /* :248 */
    out->has_local_key = p->local_key != NULL;
    out->has_layer_key = p->layer_key != NULL;
    out->opacity       = (float)(t->is_key ? 1.0 : t->opacity); /* :252 */

    if (t->chroma_enable) { /* :254-266 */
        out->chroma                           = 1;
        out->chroma_show_mask                 = t->chroma_show_mask;
        out->chroma_target_hue                = (float)(t->chroma_target_hue / 360.0);
        out->chroma_hue_width                 = (float)t->chroma_hue_width;
        out->chroma_min_saturation            = (float)t->chroma_min_saturation;
        out->chroma_min_brightness            = (float)t->chroma_min_brightness;
        out->chroma_softness                  = (float)(1.0 + t->chroma_softness);
        out->chroma_spill_suppress            = (float)(t->chroma_spill_suppress / 360.0);
        out->chroma_spill_suppress_saturation = (float)t->chroma_spill_suppress_saturation;
    }
    out->blend_mode = t->is_key ? REF_BLEND_NORMAL : p->blend_mode; /* :270-272 */
    out->keyer      = p->keyer;
    out->invert     = t->invert;

    if (t->lv_min_input > epsilon || t->lv_max_input < 1.0 - epsilon || t->lv_min_output > epsilon ||
        t->lv_max_output < 1.0 - epsilon || fabs(t->lv_gamma - 1.0) > epsilon) { /* :282-293 */
        out->levels     = 1;
        out->min_input  = (float)t->lv_min_input;
        out->max_input  = (float)t->lv_max_input;
        out->min_output = (float)t->lv_min_output;
        out->max_output = (float)t->lv_max_output;
        out->gamma      = (float)t->lv_gamma;
    }
    if (fabs(t->brightness - 1.0) > epsilon || fabs(t->saturation - 1.0) > epsilon ||
        fabs(t->contrast - 1.0) > epsilon) { /* :295-305 */
        out->csb = 1;
        out->brt = (float)t->brightness;
        out->sat = (float)t->saturation;
        out->con = (float)t->contrast;
    }

    { /* scissor :312-329 */
        const double* m_p = t->clip_translation;
        const double* m_s = t->clip_scale;
        if (m_p[0] > DBL_EPSILON || m_p[1] > DBL_EPSILON || m_s[0] < 1.0 - DBL_EPSILON ||
            m_s[1] < 1.0 - DBL_EPSILON) {
            double w     = (double)be->tex_width(p->background);
            double h     = (double)be->tex_height(p->background);
            out->scissor = 1;
            out->sc_x    = (int)(m_p[0] * w);
            out->sc_y    = (int)(m_p[1] * h);
            int sw       = (int)(m_s[0] * w);
            int sh       = (int)(m_s[1] * h);
            out->sc_w    = sw > 0 ? sw : 0;
            out->sc_h    = sh > 0 ? sh : 0;
        }
    }

    { /* perspective q :335-375 (pers[] already holds the -1 adjusted corners) */
        const double *cul = t->crop_ul, *clr = t->crop_lr;
        double        ix, iy;
        if (get_line_intersection(pers[0][0] + cul[0], pers[0][1] + cul[1], pers[2][0] + clr[0],
                                  pers[2][1] + clr[1], pers[1][0] + clr[0], pers[1][1] + cul[1],
                                  pers[3][0] + cul[0], pers[3][1] + clr[1], &ix, &iy) &&
            is_default_geometry) {
            double d0 = hypotenuse(pers[3][0] + cul[0], pers[3][1] + clr[1], ix, iy);
            double d1 = hypotenuse(pers[2][0] + clr[0], pers[2][1] + clr[1], ix, iy);
            double d2 = hypotenuse(pers[1][0] + clr[0], pers[1][1] + cul[1], ix, iy);
            double d3 = hypotenuse(pers[0][0] + cul[0], pers[0][1] + cul[1], ix, iy);
            double q[4] = {calc_q(d3, d1), calc_q(d2, d0), calc_q(d1, d3), calc_q(d0, d2)};
            for (int i = 0; i < 4; ++i) {
                coords[i].texture_q = q[i];
                coords[i].texture_x *= q[i];
                coords[i].texture_y *= q[i];
            }
        }
    }
    if (!p->is_quad) /* :378,413 only quads draw */
        return 0;
    memcpy(out->coords, coords, sizeof coords);
    out->field_mode = p->field_mode;
    return 1;
}

static void kernel_draw(ref_mixer* m, const kernel_params* p)
{
    ref_draw d;
    if (!kernel_resolve(m->be, p, &d))
        return;
    if (m->log_enabled) {
        m->log            = (ref_draw*)grow(m->log, &m->caplog, m->nlog + 1, sizeof(ref_draw));
        m->log[m->nlog++] = d;
    }
    m->be->draw(&d);
    m->draws_issued++;
}

/* ------------------------------------------------------------------ image_renderer */

typedef struct render_ctx {
    ref_mixer*        m;
    const ref_format* fmt;
    void**            temps; /* temporaries to destroy at end of render */
    int               ntemps, captemps;
} render_ctx;

static void* new_temp(render_ctx* c, int w, int h, int stride)
{
    void* t  = c->m->be->tex_create(w, h, stride, c->fmt->bits);
    c->temps = (void**)grow(c->temps, &c->captemps, c->ntemps + 1, sizeof(void*));
    c->temps[c->ntemps++] = t;
    return t;
}

/* draw(target, source_buffer, blend_mode)  image_mixer.cpp:207-225. The reference's parameter is a NAMED rvalue
 * reference: `spl::make_shared_ptr(source_buffer)` (common/memory.h:633, by value) copies from it, so the caller's
 * pointer is NOT reset. A layer's mix texture therefore stays set once created: it is composited again in front of
 * every later plain item (:197) and at the end of the layer (:138/:150), and later mix items accumulate into it. */
static void draw_buffer(render_ctx* c, void* target, void** source_buffer, int blend_mode)
{
    if (!*source_buffer)
        return;
    ref_transform def;
    ref_transform_default(&def);
    ref_coord g[4];
    ref_geometry_default(g);
    kernel_params kp;
    memset(&kp, 0, sizeof kp);
    kp.format       = REF_PF_BGRA;
    kp.nplanes      = 1;
    kp.textures[0]  = *source_buffer;
    kp.transform    = &def;
    kp.is_quad      = 1;
    kp.ncoords      = 4;
    kp.coords       = g;
    kp.blend_mode   = blend_mode;
    kp.keyer        = REF_KEYER_LINEAR;
    kp.background   = target;
    kp.aspect_ratio = 1.0; /* draw_params default :51 */
    kp.field_mode   = c->fmt->field_mode;
    kernel_draw(c->m, &kp);
}

/* draw(target, item, layer_key, local_key, local_mix)  image_mixer.cpp:156-205 */
static void draw_item(render_ctx* c, void* target, const ref_item* item, void** layer_key, void** local_key,
                      void** local_mix)
{
    const ref_backend* be = c->m->be;
    kernel_params      kp;
    memset(&kp, 0, sizeof kp);
    kp.format    = item->format;
    kp.nplanes   = item->nplanes;
    memcpy(kp.textures, item->textures, sizeof kp.textures);
    kp.transform    = &item->transform;
    kp.is_quad      = item->is_quad;
    kp.ncoords      = item->ncoords;
    kp.coords       = item->coords;
    kp.blend_mode   = REF_BLEND_NORMAL;
    kp.keyer        = REF_KEYER_LINEAR;
    kp.aspect_ratio = (double)c->fmt->square_width / (double)c->fmt->square_height; /* :167-168 */
    kp.field_mode   = c->fmt->field_mode;
    int tw = be->tex_width(target), th = be->tex_height(target);

    if (item->transform.is_key) { /* :174-183 */
        if (!*local_key)
            *local_key = new_temp(c, tw, th, 1);
        kp.background = *local_key;
        kp.local_key  = NULL;
        kp.layer_key  = NULL;
        kernel_draw(c->m, &kp);
    } else if (item->transform.is_mix) { /* :184-195 */
        if (!*local_mix)
            *local_mix = new_temp(c, tw, th, 4);
        kp.background = *local_mix;
        kp.local_key  = *local_key;
        *local_key    = NULL; /* std::move */
        kp.layer_key  = *layer_key;
        kp.keyer      = REF_KEYER_ADDITIVE;
        kernel_draw(c->m, &kp);
    } else { /* :196-204 */
        draw_buffer(c, target, local_mix, REF_BLEND_NORMAL);
        kp.background = target;
        kp.local_key  = *local_key;
        *local_key    = NULL;
        kp.layer_key  = *layer_key;
        kernel_draw(c->m, &kp);
    }
}

static void draw_layers(render_ctx* c, void* target, const ref_layer* layers, int n);

/* draw(target, layer, layer_key)  image_mixer.cpp:116-154 */
static void draw_layer(render_ctx* c, void* target, const ref_layer* layer, void** layer_key)
{
    if (layer->nitems == 0)
        return;
    void* local_key = NULL;
    void* local_mix = NULL;
    if (layer->blend_mode != REF_BLEND_NORMAL) {
        void* layer_texture = new_temp(c, c->m->be->tex_width(target), c->m->be->tex_height(target), 4);
        for (int i = 0; i < layer->nitems; ++i)
            draw_item(c, layer_texture, &layer->items[i], layer_key, &local_key, &local_mix);
        draw_buffer(c, layer_texture, &local_mix, REF_BLEND_NORMAL);
        draw_buffer(c, target, &layer_texture, layer->blend_mode);
    } else {
        for (int i = 0; i < layer->nitems; ++i)
            draw_item(c, target, &layer->items[i], layer_key, &local_key, &local_mix);
        draw_buffer(c, target, &local_mix, REF_BLEND_NORMAL);
    }
    *layer_key = local_key; /* :153 */
}

/* draw(target, layers)  image_mixer.cpp:104-114 */
static void draw_layers(render_ctx* c, void* target, const ref_layer* layers, int n)
{
    void* layer_key = NULL;
    for (int i = 0; i < n; ++i) {
        draw_layers(c, target, layers[i].sublayers, layers[i].nsub);
        draw_layer(c, target, &layers[i], &layer_key);
    }
}

/* render: image_mixer.cpp:301-304 + image_renderer::operator() :86-101.
 * out receives W*H*4 samples of fmt->bits each, memory order B,G,R,A, row 0 = top.
 * Returns number of GL draws issued (>=0). */
int ref_mixer_render(ref_mixer* m, const ref_format* fmt, void* out)
{
    size_t bytes    = (size_t)fmt->width * fmt->height * 4 * (fmt->bits / 8);
    m->draws_issued = 0;
    m->nlog         = 0;
    if (m->nlayers == 0) { /* :89-92 */
        memset(out, 0, bytes);
        return 0;
    }
    render_ctx c;
    memset(&c, 0, sizeof c);
    c.m          = m;
    c.fmt        = fmt;
    void* target = m->be->tex_create(fmt->width, fmt->height, 4, fmt->bits); /* :95 */
    draw_layers(&c, target, m->layers, m->nlayers);
    m->be->tex_read(target, out); /* :99 */
    m->be->tex_destroy(target);
    for (int i = 0; i < c.ntemps; ++i)
        m->be->tex_destroy(c.temps[i]);
    free(c.temps);
    scene_clear(m); /* layers_ is moved into the renderer :303 */
    m->npath = 0;
    return m->draws_issued;
}

void ref_mixer_enable_log(ref_mixer* m, int on) { m->log_enabled = on; }
int  ref_mixer_log_size(const ref_mixer* m) { return m->nlog; }
const ref_draw* ref_mixer_log(const ref_mixer* m) { return m->log; }
