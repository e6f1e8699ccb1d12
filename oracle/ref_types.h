/*
 * oracle/ref_types.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Plain-C restatement of the value types the reference image mixer consumes.
 * Nothing under oracle/ is linked into, imported by or called from the product
 * (server_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker.
 *
 * Reference (paths under /root/reference/src):
 *   core/frame/frame_transform.h:34-109   chroma, levels, corners, rectangle, image_transform
 *   core/frame/geometry.h:38-46           frame_geometry::coord
 *   core/frame/pixel_format.h:28-76       pixel_format, pixel_format_desc::plane
 *   core/mixer/image/blend_modes.h:28-61  blend_mode
 *   core/video_format.h                   video_format_desc (width/height/square_*)
 *   accelerator/ogl/image/image_kernel.h:34-52   keyer, draw_params
 */
#ifndef ORACLE_REF_TYPES_H
#define ORACLE_REF_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum ref_pixel_format { /* pixel_format.h:28-43 */
    REF_PF_GRAY = 0, REF_PF_BGRA, REF_PF_RGBA, REF_PF_ARGB, REF_PF_ABGR, REF_PF_YCBCR, REF_PF_YCBCRA,
    REF_PF_LUMA, REF_PF_BGR, REF_PF_RGB, REF_PF_UYVY, REF_PF_COUNT, REF_PF_INVALID
};

enum ref_keyer { REF_KEYER_LINEAR = 0, REF_KEYER_ADDITIVE = 1 }; /* image_kernel.h:34-38 */

/* blend_modes.h:28-61; only the numeric values matter to the shader. */
#define REF_BLEND_NORMAL 0
#define REF_BLEND_MIX 29

typedef struct ref_coord { /* geometry.h:38-46 */
    double vertex_x, vertex_y, texture_x, texture_y, texture_r, texture_q;
} ref_coord;

typedef struct ref_transform { /* frame_transform.h:77-99, field for field */
    double opacity, contrast, brightness, saturation;
    double anchor[2], fill_translation[2], fill_scale[2], clip_translation[2], clip_scale[2];
    double angle;
    double crop_ul[2], crop_lr[2];
    double pers_ul[2], pers_ur[2], pers_lr[2], pers_ll[2];
    double lv_min_input, lv_max_input, lv_gamma, lv_min_output, lv_max_output;
    int32_t chroma_enable, chroma_show_mask;
    double chroma_target_hue, chroma_hue_width, chroma_min_saturation, chroma_min_brightness, chroma_softness,
        chroma_spill_suppress, chroma_spill_suppress_saturation;
    int32_t is_key, invert, is_mix, blend_mode, layer_depth;
    int32_t pad_;
} ref_transform;

typedef struct ref_plane { /* pixel_format.h:47-65 (+ bits: 16-bit extension) */
    int32_t width, height, stride; /* stride = channels per texel (1..4) */
    int32_t bits;                  /* 8 (reference) or 16 (extension) */
    const void* data;              /* tightly packed rows, width*stride samples each */
} ref_plane;

typedef struct ref_frame {
    int32_t format; /* ref_pixel_format */
    int32_t nplanes;
    ref_plane planes[4];
    int32_t geometry_is_quad; /* frame_geometry::geometry_type::quad */
    int32_t ncoords;          /* 4 for a quad, 0 = empty geometry */
    ref_coord coords[4];
} ref_frame;

typedef struct ref_format { /* video_format_desc subset */
    int32_t width, height, square_width, square_height;
    int32_t bits; /* 8 (reference) or 16 (extension): depth of target + temporaries */
    int32_t field_mode; /* extension: 0 progressive, 1 = rows y%2==0 only, 2 = rows y%2==1 only */
} ref_format;

/* One draw after image_kernel::draw's host math (image_kernel.cpp:134-375): everything the GL
 * pipeline receives. Backends consume this. */
typedef struct ref_draw {
    int32_t pixel_format, nplanes;
    void* planes[4]; /* backend texture handles */
    void* background;
    void* local_key;
    void* layer_key;
    int32_t is_hd, has_local_key, has_layer_key, blend_mode, keyer, invert;
    float opacity;
    int32_t levels;
    float min_input, max_input, min_output, max_output, gamma;
    int32_t csb;
    float brt, sat, con;
    int32_t chroma, chroma_show_mask;
    float chroma_target_hue, chroma_hue_width, chroma_min_saturation, chroma_min_brightness, chroma_softness,
        chroma_spill_suppress, chroma_spill_suppress_saturation;
    int32_t scissor, sc_x, sc_y, sc_w, sc_h;
    int32_t field_mode; /* extension, see ref_format */
    ref_coord coords[4];
} ref_draw;

/* A backend executes draws on textures: the llvmpipe one (oracle/gl_backend.c, reference shaders as
 * data) or the scalar CPU restatement (oracle/ref_cpu.c). */
typedef struct ref_backend {
    void* (*tex_create)(int w, int h, int stride, int bits); /* cleared, like device.cpp:446 */
    void (*tex_upload)(void* tex, const void* data);         /* texture.cpp:85-98 */
    void (*tex_read)(void* tex, void* dst);                  /* texture.cpp:100-105 */
    void (*tex_destroy)(void* tex);
    int (*tex_width)(void* tex);
    int (*tex_height)(void* tex);
    void (*draw)(const ref_draw* d); /* image_kernel.cpp:377-419 */
} ref_backend;

#ifdef __cplusplus
}
#endif
#endif
