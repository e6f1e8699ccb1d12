/*
 * oracle/gl_backend.c -- TEST INFRASTRUCTURE ONLY. Golden-vector generator.
 *
 * Executes the REFERENCE's own GLSL (accelerator/ogl/image/shader.vert + shader.frag, loaded as data
 * from /root/reference at run time -- never copied into this repo) on a headless Mesa llvmpipe
 * GL 4.5 core context, replaying the GL call sequence of
 *   accelerator/ogl/image/image_kernel.cpp:224-419   (bind, uniforms, viewport, scissor, VBO, draw)
 *   accelerator/ogl/util/texture.cpp:31-105          (storage formats, filters, upload, readback, clear)
 *   accelerator/ogl/util/device.cpp:116-117          (one FBO bound for the device's life)
 *   accelerator/ogl/util/shader.cpp:126-134          (doubles narrowed to float on upload)
 * Only built and run in the authoring container (output: oracle/_ref/libglref.so). The GPU box has
 * no /root/reference, so nothing there depends on this file; fixtures it produced live in tests/golden/.
 *
 * 16-bit textures (GL_R16/GL_RG16/GL_RGB16/GL_RGBA16) are the north-star extension:
 * same shader, wider texels; not reference behaviour.
 *
 * Context bring-up: DRI swrast loaded directly (no X/EGL/OSMesa in the image), per SURVEY appendix A.
 * Run with env force_compat_shaders=true (shader.frag:471 uses texture2D under #version 450).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

#include "ref_types.h"

/* ---- GL entry points (all through _glapi_get_proc_address; libGL's glvnd stubs are no-ops here) ---- */
#define GL_FUNCS                                                                                                      \
    X(PFNGLGETSTRINGPROC, glGetString)                                                                                 \
    X(PFNGLGETERRORPROC, glGetError)                                                                                   \
    X(PFNGLCREATETEXTURESPROC, glCreateTextures)                                                                       \
    X(PFNGLDELETETEXTURESPROC, glDeleteTextures)                                                                       \
    X(PFNGLTEXTUREPARAMETERIPROC, glTextureParameteri)                                                                 \
    X(PFNGLTEXTURESTORAGE2DPROC, glTextureStorage2D)                                                                   \
    X(PFNGLTEXTURESUBIMAGE2DPROC, glTextureSubImage2D)                                                                 \
    X(PFNGLCLEARTEXIMAGEPROC, glClearTexImage)                                                                         \
    X(PFNGLGETTEXTUREIMAGEPROC, glGetTextureImage)                                                                     \
    X(PFNGLPIXELSTOREIPROC, glPixelStorei)                                                                             \
    X(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers)                                                                     \
    X(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer)                                                                     \
    X(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D)                                                           \
    X(PFNGLACTIVETEXTUREPROC, glActiveTexture)                                                                         \
    X(PFNGLBINDTEXTUREPROC, glBindTexture)                                                                             \
    X(PFNGLCREATESHADERPROC, glCreateShader)                                                                           \
    X(PFNGLSHADERSOURCEPROC, glShaderSource)                                                                           \
    X(PFNGLCOMPILESHADERPROC, glCompileShader)                                                                         \
    X(PFNGLGETSHADERIVPROC, glGetShaderiv)                                                                             \
    X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog)                                                                   \
    X(PFNGLCREATEPROGRAMPROC, glCreateProgram)                                                                         \
    X(PFNGLATTACHSHADERPROC, glAttachShader)                                                                           \
    X(PFNGLLINKPROGRAMPROC, glLinkProgram)                                                                             \
    X(PFNGLGETPROGRAMIVPROC, glGetProgramiv)                                                                           \
    X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog)                                                                 \
    X(PFNGLUSEPROGRAMPROC, glUseProgram)                                                                               \
    X(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation)                                                               \
    X(PFNGLGETATTRIBLOCATIONPROC, glGetAttribLocation)                                                                 \
    X(PFNGLUNIFORM1IPROC, glUniform1i)                                                                                 \
    X(PFNGLUNIFORM1FPROC, glUniform1f)                                                                                 \
    X(PFNGLVIEWPORTPROC, glViewport)                                                                                   \
    X(PFNGLSCISSORPROC, glScissor)                                                                                     \
    X(PFNGLENABLEPROC, glEnable)                                                                                       \
    X(PFNGLDISABLEPROC, glDisable)                                                                                     \
    X(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays)                                                                     \
    X(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray)                                                                     \
    X(PFNGLGENBUFFERSPROC, glGenBuffers)                                                                               \
    X(PFNGLBINDBUFFERPROC, glBindBuffer)                                                                               \
    X(PFNGLBUFFERDATAPROC, glBufferData)                                                                               \
    X(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray)                                                     \
    X(PFNGLDISABLEVERTEXATTRIBARRAYPROC, glDisableVertexAttribArray)                                                   \
    X(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer)                                                             \
    X(PFNGLDRAWARRAYSPROC, glDrawArrays)                                                                               \
    X(PFNGLTEXTUREBARRIERPROC, glTextureBarrier)                                                                       \
    X(PFNGLFINISHPROC, glFinish)

#define X(type, name) static type p_##name;
GL_FUNCS
#undef X

/* ---- DRI swrast loader stubs ---- */
static void gdi(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* p)
{
    (void)d; (void)p;
    *x = *y = 0;
    *w = *h = 16;
}
static void pi(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* p)
{
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void gi(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* p)
{
    (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void pi2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* p)
{
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}
static void gi2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* p)
{
    (void)d; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}
static const __DRIswrastLoaderExtension swl = {.base            = {__DRI_SWRAST_LOADER, 3},
                                               .getDrawableInfo = gdi,
                                               .putImage        = pi,
                                               .getImage        = gi,
                                               .putImage2       = pi2,
                                               .getImage2       = gi2};
static const __DRIextension*            loader_exts[] = {&swl.base, NULL};

typedef struct gl_tex {
    GLuint id;
    int    w, h, stride, bits;
} gl_tex;

static int    g_ready = 0;
static GLuint g_prog, g_fbo, g_vao, g_vbo;
static char   g_info[512];

static char* slurp(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f)
        return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* s = (char*)malloc((size_t)n + 1);
    if (fread(s, 1, (size_t)n, f) != (size_t)n) {
        fclose(f);
        free(s);
        return NULL;
    }
    s[n] = 0;
    fclose(f);
    return s;
}

static GLuint compile(GLenum type, const char* src)
{
    GLuint s = p_glCreateShader(type);
    p_glShaderSource(s, 1, &src, NULL);
    p_glCompileShader(s);
    GLint ok = 0;
    p_glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[4096];
        p_glGetShaderInfoLog(s, sizeof log, NULL, log);
        fprintf(stderr, "glref: shader compile failed:\n%s\n", log);
        return 0;
    }
    return s;
}

/* Returns 0 on success. reference_dir = ".../src/accelerator/ogl/image" */
int glref_init(const char* reference_dir)
{
    if (g_ready)
        return 0;
    setenv("force_compat_shaders", "true", 1);
    void* drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) {
        fprintf(stderr, "glref: %s\n", dlerror());
        return -1;
    }
    const __DRIextension** (*getext)(void) =
        (const __DRIextension** (*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!getext)
        return -2;
    const __DRIextension**     ext  = getext();
    const __DRIcoreExtension*  core = NULL;
    const __DRIswrastExtension* sw  = NULL;
    for (int i = 0; ext[i]; ++i) {
        if (!strcmp(ext[i]->name, __DRI_CORE))
            core = (const __DRIcoreExtension*)ext[i];
        if (!strcmp(ext[i]->name, __DRI_SWRAST))
            sw = (const __DRIswrastExtension*)ext[i];
    }
    if (!core || !sw)
        return -3;
    const __DRIconfig** cfg = NULL;
    __DRIscreen*        scr = sw->createNewScreen2(0, loader_exts, ext, &cfg, NULL);
    if (!scr)
        return -4;
    uint32_t       at[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5};
    unsigned       err  = 0;
    __DRIcontext*  ctx  = sw->createContextAttribs(scr, __DRI_API_OPENGL_CORE, cfg[0], NULL, 2, at, &err, NULL);
    if (!ctx)
        return -5;
    __DRIdrawable* dr = sw->createNewDrawable(scr, cfg[0], NULL);
    if (!core->bindContext(ctx, dr, dr))
        return -6;
    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi)
        return -7;
    void* (*gp)(const char*) = (void* (*)(const char*))dlsym(glapi, "_glapi_get_proc_address");
    if (!gp)
        return -8;
#define X(type, name)                                                                                                  \
    p_##name = (type)gp(#name);                                                                                        \
    if (!p_##name) {                                                                                                   \
        fprintf(stderr, "glref: missing %s\n", #name);                                                                 \
        return -9;                                                                                                     \
    }
    GL_FUNCS
#undef X
    snprintf(g_info, sizeof g_info, "%s | %s", (const char*)p_glGetString(GL_VERSION),
             (const char*)p_glGetString(GL_RENDERER));

    char path[1024];
    snprintf(path, sizeof path, "%s/shader.vert", reference_dir);
    char* vs = slurp(path);
    snprintf(path, sizeof path, "%s/shader.frag", reference_dir);
    char* fs = slurp(path);
    if (!vs || !fs) {
        fprintf(stderr, "glref: cannot read reference shaders under %s\n", reference_dir);
        return -10;
    }
    GLuint v = compile(GL_VERTEX_SHADER, vs), f = compile(GL_FRAGMENT_SHADER, fs);
    free(vs);
    free(fs);
    if (!v || !f)
        return -11;
    g_prog = p_glCreateProgram();
    p_glAttachShader(g_prog, v);
    p_glAttachShader(g_prog, f);
    p_glLinkProgram(g_prog);
    GLint ok = 0;
    p_glGetProgramiv(g_prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[4096];
        p_glGetProgramInfoLog(g_prog, sizeof log, NULL, log);
        fprintf(stderr, "glref: link failed:\n%s\n", log);
        return -12;
    }
    p_glGenFramebuffers(1, &g_fbo); /* device.cpp:116-117 */
    p_glBindFramebuffer(GL_FRAMEBUFFER, g_fbo);
    p_glGenVertexArrays(1, &g_vao); /* image_kernel.cpp:121-122 */
    p_glGenBuffers(1, &g_vbo);
    g_ready = 1;
    return (int)p_glGetError();
}

const char* glref_info(void) { return g_info; }

/* texture.cpp:31-33 (+16-bit rows) */
static const GLenum FORMAT[]    = {0, GL_RED, GL_RG, GL_BGR, GL_BGRA};
static const GLenum INTERNAL8[] = {0, GL_R8, GL_RG8, GL_RGB8, GL_RGBA8};
static const GLenum INTERNAL16[] = {0, GL_R16, GL_RG16, GL_RGB16, GL_RGBA16};
static const GLenum TYPE8[] = {0, GL_UNSIGNED_BYTE, GL_UNSIGNED_BYTE, GL_UNSIGNED_BYTE, GL_UNSIGNED_INT_8_8_8_8_REV};

static GLenum tex_type(const gl_tex* t) { return t->bits == 16 ? GL_UNSIGNED_SHORT : TYPE8[t->stride]; }

static void gl_tex_clear(gl_tex* t) { p_glClearTexImage(t->id, 0, FORMAT[t->stride], tex_type(t), NULL); }

static void* gl_tex_create(int w, int h, int stride, int bits)
{
    gl_tex* t = (gl_tex*)calloc(1, sizeof *t);
    t->w = w, t->h = h, t->stride = stride, t->bits = bits;
    p_glCreateTextures(GL_TEXTURE_2D, 1, &t->id); /* texture.cpp:53-58 */
    p_glTextureParameteri(t->id, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    p_glTextureParameteri(t->id, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    p_glTextureParameteri(t->id, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    p_glTextureParameteri(t->id, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    p_glTextureStorage2D(t->id, 1, bits == 16 ? INTERNAL16[stride] : INTERNAL8[stride], w, h);
    gl_tex_clear(t); /* device.cpp:207-209,446 */
    return t;
}

static void gl_tex_upload(void* tex, const void* data)
{
    gl_tex* t = (gl_tex*)tex;
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, (t->w % 16 > 0) ? 1 : 4); /* texture.cpp:89-93 */
    p_glTextureSubImage2D(t->id, 0, 0, 0, t->w, t->h, FORMAT[t->stride], tex_type(t), data);
}

static void gl_tex_read(void* tex, void* dst)
{
    gl_tex* t = (gl_tex*)tex;
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glGetTextureImage(t->id, 0, FORMAT[t->stride], tex_type(t), t->w * t->h * t->stride * (t->bits / 8), dst);
    p_glFinish();
}

static void gl_tex_destroy(void* tex)
{
    gl_tex* t = (gl_tex*)tex;
    p_glDeleteTextures(1, &t->id);
    free(t);
}
static int gl_tex_width(void* tex) { return ((gl_tex*)tex)->w; }
static int gl_tex_height(void* tex) { return ((gl_tex*)tex)->h; }

static void bind_tex(const void* tex, int unit)
{ /* texture.cpp:63-69 */
    p_glActiveTexture(GL_TEXTURE0 + unit);
    p_glBindTexture(GL_TEXTURE_2D, ((const gl_tex*)tex)->id);
}
static void set_i(const char* n, int v) { p_glUniform1i(p_glGetUniformLocation(g_prog, n), v); }
static void set_f(const char* n, float v) { p_glUniform1f(p_glGetUniformLocation(g_prog, n), v); }

static void gl_draw(const ref_draw* d)
{
    /* image_kernel.cpp:224-236: texture units per image_shader.h:31-40 */
    for (int n = 0; n < d->nplanes; ++n)
        bind_tex(d->planes[n], n);
    if (d->local_key)
        bind_tex(d->local_key, 4);
    if (d->layer_key)
        bind_tex(d->layer_key, 5);

    p_glUseProgram(g_prog); /* :240-266 */
    set_i("plane[0]", 0);
    set_i("plane[1]", 1);
    set_i("plane[2]", 2);
    set_i("plane[3]", 3);
    set_i("local_key", 4);
    set_i("layer_key", 5);
    set_i("is_hd", d->is_hd);
    set_i("has_local_key", d->local_key != NULL);
    set_i("has_layer_key", d->layer_key != NULL);
    set_i("pixel_format", d->pixel_format);
    set_f("opacity", d->opacity);
    if (d->chroma) {
        set_i("chroma", 1);
        set_i("chroma_show_mask", d->chroma_show_mask);
        set_f("chroma_target_hue", d->chroma_target_hue);
        set_f("chroma_hue_width", d->chroma_hue_width);
        set_f("chroma_min_saturation", d->chroma_min_saturation);
        set_f("chroma_min_brightness", d->chroma_min_brightness);
        set_f("chroma_softness", d->chroma_softness);
        set_f("chroma_spill_suppress", d->chroma_spill_suppress);
        set_f("chroma_spill_suppress_saturation", d->chroma_spill_suppress_saturation);
    } else {
        set_i("chroma", 0);
    }
    bind_tex(d->background, 6); /* :274-277 */
    set_i("background", 6);
    set_i("blend_mode", d->blend_mode);
    set_i("keyer", d->keyer);
    set_i("invert", d->invert); /* :280-305 */
    if (d->levels) {
        set_i("levels", 1);
        set_f("min_input", d->min_input);
        set_f("max_input", d->max_input);
        set_f("min_output", d->min_output);
        set_f("max_output", d->max_output);
        set_f("gamma", d->gamma);
    } else {
        set_i("levels", 0);
    }
    if (d->csb) {
        set_i("csb", 1);
        set_f("brt", d->brt);
        set_f("sat", d->sat);
        set_f("con", d->con);
    } else {
        set_i("csb", 0);
    }
    const gl_tex* bg = (const gl_tex*)d->background;
    p_glViewport(0, 0, bg->w, bg->h); /* :309-329 */
    p_glDisable(GL_DEPTH_TEST);
    if (d->scissor) {
        p_glEnable(GL_SCISSOR_TEST);
        p_glScissor(d->sc_x, d->sc_y, d->sc_w, d->sc_h);
    }
    p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, bg->id, 0); /* :332 */

    /* :378-412 */
    p_glBindVertexArray(g_vao);
    p_glBindBuffer(GL_ARRAY_BUFFER, g_vbo);
    ref_coord tri[6] = {d->coords[0], d->coords[1], d->coords[2], d->coords[0], d->coords[2], d->coords[3]};
    p_glBufferData(GL_ARRAY_BUFFER, sizeof tri, tri, GL_STATIC_DRAW);
    GLint vtx = p_glGetAttribLocation(g_prog, "Position");
    GLint tex = p_glGetAttribLocation(g_prog, "TexCoordIn");
    p_glEnableVertexAttribArray((GLuint)vtx);
    p_glEnableVertexAttribArray((GLuint)tex);
    p_glVertexAttribPointer((GLuint)vtx, 2, GL_DOUBLE, GL_FALSE, sizeof(ref_coord), NULL);
    p_glVertexAttribPointer((GLuint)tex, 4, GL_DOUBLE, GL_FALSE, sizeof(ref_coord), (void*)(2 * sizeof(double)));
    p_glDrawArrays(GL_TRIANGLES, 0, 6);
    p_glTextureBarrier();
    p_glDisableVertexAttribArray((GLuint)vtx);
    p_glDisableVertexAttribArray((GLuint)tex);
    p_glBindVertexArray(0);
    p_glBindBuffer(GL_ARRAY_BUFFER, 0);
    p_glDisable(GL_SCISSOR_TEST); /* :418-419 */
    p_glDisable(GL_BLEND);
}

int glref_error(void) { return g_ready ? (int)p_glGetError() : -1; }

static const ref_backend g_backend = {gl_tex_create, gl_tex_upload, gl_tex_read, gl_tex_destroy,
                                      gl_tex_width,  gl_tex_height, gl_draw};

const ref_backend* glref_backend(void) { return &g_backend; }
