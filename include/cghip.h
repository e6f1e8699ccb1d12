/*
 * cghip.h -- C ABI of the MI355X-native CasparCG image mixer (libcghip.so).
 *
 * This is the drop-in boundary for the reference's `accelerator/ogl` module. A thin C++ adapter
 * (`accelerator/hip/image_mixer.cpp`, shown in INTEGRATION.md; standalone mirror in server_amd/host/)
 * implements `core::image_mixer` (core/mixer/image/image_mixer.h:33-57) on top of these entry points.
 * Plain pointers and sizes only; no C++/torch types cross it. Every function returns 0 (CGHIP_OK) or a
 * negative cghip_status; cghip_last_error() gives the thread-local message. All functions are
 * thread-safe except calls on one cghip_mixer, which are single-threaded by the reference's own
 * contract (push/visit/pop/render only from the channel thread, core/mixer/mixer.cpp:62-92).
 *
 * Reference paths are under /root/reference/src.
 */
#ifndef CGHIP_H
#define CGHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGHIP_ABI_VERSION 1

typedef enum cghip_status {
    CGHIP_OK              = 0,
    CGHIP_ERR_INVALID     = -1, /* bad argument (reference: CASPAR_VERIFY / invalid_argument) */
    CGHIP_ERR_NO_DEVICE   = -2, /* no usable gfx950 device (reference: not_supported, device.cpp:109-114) */
    CGHIP_ERR_HIP         = -3, /* a HIP runtime call failed; message holds hipGetErrorString */
    CGHIP_ERR_NOMEM       = -4,
    CGHIP_ERR_UNSUPPORTED = -5  /* scene needs more live temporaries than the fused kernel holds */
} cghip_status;

typedef struct cghip_device  cghip_device;  /* replaces ogl::device            (ogl/util/device.h) */
typedef struct cghip_hostbuf cghip_hostbuf; /* replaces ogl::buffer (pinned)   (ogl/util/buffer.h) */
typedef struct cghip_plane   cghip_plane;   /* replaces ogl::texture (sources) (ogl/util/texture.h) */
typedef struct cghip_mixer   cghip_mixer;   /* replaces image_renderer+kernel  (ogl/image/image_mixer.cpp:74-226) */
typedef struct cghip_ticket  cghip_ticket;  /* replaces future<array<const uint8_t>> of device::copy_async */

/* core/frame/pixel_format.h:28-43 */
typedef enum cghip_pixel_format {
    CGHIP_PF_GRAY = 0, CGHIP_PF_BGRA, CGHIP_PF_RGBA, CGHIP_PF_ARGB, CGHIP_PF_ABGR, CGHIP_PF_YCBCR,
    CGHIP_PF_YCBCRA, CGHIP_PF_LUMA, CGHIP_PF_BGR, CGHIP_PF_RGB, CGHIP_PF_UYVY, CGHIP_PF_COUNT, CGHIP_PF_INVALID
} cghip_pixel_format;

/* accelerator/ogl/image/image_kernel.h:34-38 */
typedef enum cghip_keyer { CGHIP_KEYER_LINEAR = 0, CGHIP_KEYER_ADDITIVE = 1 } cghip_keyer;

/* core/mixer/image/blend_modes.h:28-61 -- numeric values are passed through unchanged */
#define CGHIP_BLEND_NORMAL 0
#define CGHIP_BLEND_MIX 29
#define CGHIP_BLEND_COUNT 30

/* core/frame/geometry.h:38-46 */
typedef struct cghip_coord {
    double vertex_x, vertex_y, texture_x, texture_y, texture_r, texture_q;
} cghip_coord;

/* core/frame/frame_transform.h:77-99 (image_transform, with chroma/levels/corners/rectangle inlined) */
typedef struct cghip_transform {
    double  opacity, contrast, brightness, saturation;
    double  anchor[2], fill_translation[2], fill_scale[2], clip_translation[2], clip_scale[2];
    double  angle;
    double  crop_ul[2], crop_lr[2];
    double  pers_ul[2], pers_ur[2], pers_lr[2], pers_ll[2];
    double  levels_min_input, levels_max_input, levels_gamma, levels_min_output, levels_max_output;
    int32_t chroma_enable, chroma_show_mask;
    double  chroma_target_hue, chroma_hue_width, chroma_min_saturation, chroma_min_brightness, chroma_softness,
        chroma_spill_suppress, chroma_spill_suppress_saturation;
    int32_t is_key, invert, is_mix, blend_mode, layer_depth;
    int32_t reserved_;
} cghip_transform;

/* core/video_format.h (fields the mixer reads) + the two north-star extensions */
typedef struct cghip_format {
    int32_t width, height;               /* video_format_desc::width/height */
    int32_t square_width, square_height; /* aspect_ratio source, image_mixer.cpp:167-168 */
    int32_t bits;                        /* 8 = reference (BGRA8 out); 16 = extension (BGRA16 out) */
    int32_t field_mode;                  /* extension: 0 progressive; 1 / 2 = only even / odd rows are drawn */
} cghip_format;

/* A render surface = one GL texture the reference renderer draws INTO: slot 0 is always the W x H
 * 4-channel frame target (image_mixer.cpp:95); other slots are the temporaries of
 * image_mixer.cpp:128 (layer), :177 (local key, 1 channel), :187 (mix). Each is cleared when created
 * (device.cpp:207-209,446). The fused kernel keeps them in registers; they never touch HBM. */
typedef struct cghip_surface {
    int32_t channels; /* 4 or 1 */
    int32_t reserved_;
} cghip_surface;

/* POD of ogl::draw_params (image_kernel.h:40-52): one call of image_kernel::draw. */
typedef struct cghip_draw {
    int32_t pixel_format; /* cghip_pixel_format of the source */
    int32_t nplanes;      /* pix_desc.planes.size(); 0 when the source is a surface */
    const cghip_plane* planes[4];
    int32_t source_surface; /* >= 0: source is that surface read 1:1 as bgra (image_mixer.cpp:207-225); else -1 */
    int32_t target;         /* surface slot drawn into (draw_params::background) */
    int32_t local_key;      /* surface slot or -1 */
    int32_t layer_key;      /* surface slot or -1 */
    int32_t blend_mode;     /* draw_params::blend_mode */
    int32_t keyer;          /* cghip_keyer */
    int32_t geometry_is_quad; /* frame_geometry::geometry_type::quad */
    int32_t ncoords;          /* 4, or 0 for empty geometry (image_kernel.cpp:150) */
    cghip_coord coords[4];
    cghip_transform transform;
    double  aspect_ratio;
} cghip_draw;

#define CGHIP_RENDER_READBACK 0x1u /* copy the frame to a pinned host buffer (device::copy_async(texture)) */
/* With field_mode 1 / 2 and CGHIP_RENDER_READBACK: transfer only the rows of that field (half the D2H bytes). The
 * host buffer keeps the full-frame layout; rows of the other field are unspecified. For consumers that weave two
 * field renders themselves (modules/decklink/consumer/decklink_consumer.cpp:293-308,
 * modules/newtek/consumer/newtek_ndi_consumer.cpp:135-140). */
#define CGHIP_RENDER_FIELD_ROWS_ONLY 0x2u
/* No hipEvent pair around the frame's launches: cghip_ticket_kernel_ms reports -1 ("not measured", still CGHIP_OK). A
 * channel that does not feed a diagnostics graph saves one runtime call and one queue packet per frame. */
#define CGHIP_RENDER_UNTIMED 0x4u
/* Surface slots one frame's draw list may name (slot 0 = the frame; every key / mix / layer temporary the renderer
 * creates is one more). Slots are bound to the kernel's register surfaces by liveness, so this only bounds the list. */
#define CGHIP_MAX_SURFACES 1024

/* ---- error reporting ------------------------------------------------------------------------- */
const char* cghip_last_error(void);
int         cghip_abi_version(void);

/* ---- device: replaces ogl::device ctor/info/gc (ogl/util/device.cpp:82-142,335-436) and
 *      accelerator::get_device (accelerator/accelerator.cpp:27-34) ----------------------------- */
int cghip_device_count(int* count);
int cghip_device_create(int gpu_index, cghip_device** out);
int cghip_device_destroy(cghip_device* dev);
int cghip_device_info(cghip_device* dev, char* json, size_t capacity); /* accelerator_device::info, GL INFO */
int cghip_device_gc(cghip_device* dev);                               /* accelerator_device::gc,   GL GC   */
int cghip_device_synchronize(cghip_device* dev);

/* ---- pinned host staging: replaces device::create_array / create_buffer (device.cpp:214-240) --- */
int cghip_host_alloc(cghip_device* dev, size_t bytes, int write, void** ptr, cghip_hostbuf** out);
int cghip_host_release(cghip_hostbuf* buf);

/* ---- source planes: replaces device::copy_async(array -> texture) (device.cpp:242-262) and
 *      texture::copy_from (texture.cpp:85-98). Rows are tightly packed: width*channels*(bits/8). --- */
int cghip_plane_upload_async(cghip_device* dev, const void* src, int width, int height, int channels, int bits,
                             cghip_plane** out);
int cghip_plane_retain(cghip_plane* plane);
int cghip_plane_release(cghip_plane* plane);
int cghip_plane_desc(const cghip_plane* plane, int* width, int* height, int* channels, int* bits, int* gpu_index);
/* route producers (core/producer/route/route_producer.cpp:40-126): a plane uploaded on another GPU is
 * copied device-to-device over xGMI; the reference shares one GL context instead (accelerator.cpp:27-34).
 * The copy is enqueued, not waited for; the new plane retains `src` until it is released itself. Same device:
 * `*out` is `src`, retained. */
int cghip_plane_peer_copy_async(cghip_device* dst_dev, cghip_plane* src, cghip_plane** out);
/* A second plane over the SAME device bytes with another shape (width*height*channels must match). The uyvy
 * pixel format is two views of one buffer, RG8 w x h and RGBA8 (w/2) x h; the reference uploads those bytes
 * twice (modules/ffmpeg/util/av_util.cpp:163-171, TODO at :45). The view retains its parent. */
int cghip_plane_view(cghip_plane* src, int width, int height, int channels, cghip_plane** out);

/* ---- mixer: replaces image_renderer::operator() + image_kernel::draw
 *      (ogl/image/image_mixer.cpp:86-226, ogl/image/image_kernel.cpp:134-420) ------------------- */
int cghip_mixer_create(cghip_device* dev, int channel_id, cghip_mixer** out);
int cghip_mixer_destroy(cghip_mixer* mixer);
/* Renders one frame: `draws` is the exact sequence of kernel_.draw() calls the reference renderer would
 * issue, in order. ndraws == 0 yields the all-zero frame (image_mixer.cpp:89-92). */
int cghip_mixer_render_async(cghip_mixer* mixer, const cghip_format* format, const cghip_surface* surfaces,
                             int nsurfaces, const cghip_draw* draws, int ndraws, unsigned flags,
                             cghip_ticket** out);

/* ---- the entry point that allocates nothing. It has NO counterpart in the reference (whose renderer creates
 *      textures and buffers per frame, image_mixer.cpp:95,128,177,187). It is for callers that cannot allocate: code
 *      under HIP graph capture set-up, tight serving loops, harnesses that forbid hipMalloc. Same frame, same kernels
 *      and the same result as cghip_mixer_render_async, but
 *        - the frame is written to `device_out` (W*H*4*(bits/8) bytes of device memory on the mixer's GPU, the
 *          caller's), and the call returns when it is complete (it blocks; no ticket, no readback);
 *        - what the kernels need beyond sources and output - the op list and the surfaces a frame that is cut into
 *          several launches hands from one launch to the next - lives in a per-GPU static device array of
 *          CGHIP_NOALLOC_SCRATCH_BYTES, held until the process ends. A frame whose op list is longer than
 *          CGHIP_NOALLOC_MAX_OPS or whose hand-over surfaces need more than the rest of the array is refused with
 *          CGHIP_ERR_UNSUPPORTED (a 2160p 16-bit stack that is ONE launch, like BASELINE config 3, needs none);
 *        - no hipMalloc / hipFree / hipHostMalloc / event or stream creation happens inside the call.
 *      Every call on a GPU shares that GPU's array, which keeps the last call's contents; nothing in it has to start
 *      at zero (a launch only loads surfaces an earlier launch of the same frame stored). Calls on one GPU are
 *      serialised by the library; render_async frames do not use the array. ------------------------------------- */
#define CGHIP_NOALLOC_MAX_OPS 256
#define CGHIP_NOALLOC_SCRATCH_BYTES (48u << 20)
int cghip_mixer_render_noalloc(cghip_mixer* mixer, const cghip_format* format, const cghip_surface* surfaces,
                               int nsurfaces, const cghip_draw* draws, int ndraws, void* device_out,
                               size_t device_out_bytes);

/* ---- tickets: replaces future<array<const uint8_t>>::get (common/future.h:9-27) --------------- */
int cghip_ticket_wait(cghip_ticket* ticket, const uint8_t** data, size_t* size); /* needs CGHIP_RENDER_READBACK */
/* The frame as the GPU sees it. Normally the frame buffer in HBM; for a frame rendered with CGHIP_RENDER_READBACK as one
 * launch it is the pinned host buffer itself (direct readback: the launch stores across PCIe) -
 * device-accessible, but reads from it cross PCIe. Render without CGHIP_RENDER_READBACK, or with
 * CGHIP_READBACK_DIRECT=0 in the environment, for a frame that a GPU consumer reads. */
int cghip_ticket_device_ptr(cghip_ticket* ticket, const void** device_ptr, size_t* size);
/* hipEvent time of the frame's launches in ms; -1 for a frame rendered with CGHIP_RENDER_UNTIMED */
int cghip_ticket_kernel_ms(cghip_ticket* ticket, float* ms, int* launches);
int cghip_ticket_release(cghip_ticket* ticket);

#ifdef __cplusplus
}
#endif
#endif /* CGHIP_H */
