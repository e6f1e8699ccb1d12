/* cghip_internal.h -- private declarations shared by the libcghip.so translation units. */
#ifndef CGHIP_INTERNAL_H
#define CGHIP_INTERNAL_H

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cghip.h"
#include "cghip_kops.h"

namespace cghip {

void set_error(const std::string& msg);
int  fail(int status, const std::string& msg);
int  fail_hip(hipError_t e, const char* what);

#define CGHIP_TRY_HIP(expr)                                                                                            \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return cghip::fail_hip(e__, #expr);                                                                        \
    } while (0)

/* size-keyed free lists, the HIP counterpart of ogl::device's texture / host-buffer pools
 * (accelerator/ogl/util/device.cpp:66-67,194-233) */
struct pool {
    std::multimap<size_t, void*> free_;
    size_t                       bytes_free = 0, bytes_total = 0, count_total = 0;
};

} // namespace cghip

struct cghip_device {
    int         gpu = 0;
    std::string name;
    hipStream_t upload_stream = nullptr; /* H2D of producer planes; shared by all producer threads */
    hipStream_t peer_stream   = nullptr; /* device-to-device copies of routed planes INTO this device: a 16.6 MB plane
                                            arriving over xGMI must not queue behind this GPU's PCIe uploads */
    /* peer access from this GPU to the memory of GPU k, asked for lazily at the first routed plane from k
       (hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess): 0 not asked, 1 granted (copies go device to device over
       xGMI), -1 refused (the runtime stages the copy through host memory), 2 the same physical GPU */
    std::map<int, int> peer_access;
    std::mutex  mu;          /* peer_access and the counters' snapshot */
    std::mutex  mu_dev, mu_pin, mu_ev; /* one lock per pool: 32 channel threads and the producers' upload threads take
                                          device blocks, pinned blocks and events independently */
    cghip::pool device_pool; /* hipMalloc'd blocks: planes, frame outputs, op lists */
    cghip::pool pinned_pool; /* hipHostMalloc'd blocks */
    /* hipEvents are pooled like memory: a frame used to create and destroy three or four of them. EV_TIMED: start /
       stop of the compositor (elapsed time is read); EV_DONE: completion a thread sleeps on (blocking sync instead of
       the default busy wait: 32 channel threads of one GPU would each spin a core); EV_ORDER: stream-to-stream order */
    enum { EV_TIMED = 0, EV_DONE = 1, EV_ORDER = 2, EV_KINDS = 3 };
    std::vector<hipEvent_t> event_pool[EV_KINDS];
    std::atomic<int64_t>    events_created{0};
    int  event_get(int kind, hipEvent_t* out);
    void event_put(int kind, hipEvent_t ev);
    std::atomic<int64_t> planes_live{0}, tickets_live{0}, frames_rendered{0}, kernel_launches{0}, mixers{0};
    std::atomic<int64_t> peer_copies{0}, peer_copy_bytes{0}; /* routed planes copied INTO this device */
    std::atomic<int64_t> uploads{0}, upload_bytes{0};        /* host-to-device plane uploads */

    /* cghip_mixer_render_noalloc: pinned staging for a frame's op list and op masks, taken ONCE here at device set-up
       (cghip_device_create), so that the call itself neither allocates nor makes the runtime stage or pin a pageable
       source per call. Used under the GPU's noalloc lock. */
    void*  noalloc_staging       = nullptr;
    size_t noalloc_staging_bytes = 0;

    int  dev_alloc(size_t bytes, void** out);
    void dev_free(void* p, size_t bytes);
    int  pinned_alloc(size_t bytes, void** out);
    void pinned_free(void* p, size_t bytes);
};

struct cghip_hostbuf {
    cghip_device* dev;
    void*         ptr;
    size_t        bytes;
};

struct cghip_plane {
    cghip_device*    dev      = nullptr;
    void*            dptr     = nullptr;
    size_t           bytes    = 0;
    int              width = 0, height = 0, channels = 0, bits = 0;
    hipEvent_t       ready    = nullptr; /* recorded after the upload / peer copy */
    std::atomic<bool> ready_seen{false}; /* the event was once found complete: later frames need not wait on it */
    std::atomic<int> refs{1};
    bool             stub = false; /* planner tests: no device memory behind it */
    cghip_plane*     parent = nullptr; /* cghip_plane_view: dptr and ready belong to the parent, which is retained */
    cghip_plane*     origin = nullptr; /* peer copy: the source plane, retained until this copy is released */
};

struct cghip_mixer {
    cghip_device* dev        = nullptr;
    int           channel_id = 0;
    hipStream_t   stream     = nullptr; /* compositor of this channel */
    unsigned      direct_frames = 0; /* direct readback: which of the two streams the next such frame takes */
    hipStream_t   ops_stream = nullptr; /* uploads a frame's op list while the previous frame is still compositing */
    hipStream_t   readback_stream = nullptr; /* D2H of finished frames (device.cpp:264-302): frame n crosses PCIe while
                                                frame n + 1 composites; on the compositor's stream its 33 - 66 MB copy
                                                (0.65 - 1.3 ms) sat in front of the next 0.2 ms launch */
};

struct cghip_ticket {
    cghip_device*             dev   = nullptr;
    hipStream_t               stream = nullptr;
    hipStream_t               readback_stream = nullptr; /* set when the frame's D2H was enqueued there */
    void*                     d_out = nullptr;
    size_t                    out_bytes = 0;
    void*                     h_out = nullptr; /* pinned, when CGHIP_RENDER_READBACK */
    void*                     d_ops = nullptr;
    void*                     h_ops = nullptr; /* pinned staging of the op list */
    size_t                    ops_bytes = 0;
    hipEvent_t                ev_start = nullptr, ev_stop = nullptr, ev_done = nullptr, ev_ops = nullptr;
    int                       launches = 0;
    void*                     spill[2] = {nullptr, nullptr}; /* RGBA1, RGBA2 between launches */
    size_t                    spill_bytes[2] = {0, 0};
    std::vector<void*>        key_bufs; /* key surfaces between launches (frame_plan::n_key_bufs) */
    size_t                    key_buf_bytes = 0;
    std::vector<cghip_plane*> planes; /* retained until the frame is done */
};

namespace cghip {

/* One kernel launch of a frame: a run of consecutive ops that one kernel class executes (cghip_plan.cpp). */
struct segment {
    int      first = 0, count = 0; /* ops[first, first + count) */
    bool     fast = false;         /* cghip_composite_fast, else cghip_composite */
    bool     full = false;         /* touches RGBA1 / RGBA2 */
    bool     fused = false;        /* holds a KOp::fused op */
    int      one_temp = 0;         /* full, but RGBA2 is never touched, loaded or stored: 1; and no op with a blend mode
                                      (a MIX transition outside blend-mode layers): 2 */
    bool     mix_pairs = false;    /* holds fused MIX pairs (BB_MIX_HEAD) */
    bool     yuv_lut = false;      /* has aligned draws of 8-bit planar / uyvy sources (KSpill::yuv_lut) */
    int      ext = 0;              /* fast: 1 a draw takes a sampled fetch body (FB_RECT8 ...), 2 a fused blend-mode layer */
    int      mono = -1;            /* fast, not full, ext 0: the one fetch body all its draws take (KSpill::mono), else -1 */
    uint32_t load_mask = 0, store_mask = 1; /* KSpill masks */
    /* key buffers (frame_plan::n_key_bufs of them, W x H samples each) the launch loads KEY0 / KEY1 from and stores them
       to, -1 = none: a key texture parked in HBM keeps its buffer while its register serves another (cghip_plan.cpp) */
    int      key_ld[2] = {-1, -1}, key_st[2] = {-1, -1};
    int      gx0 = 0, gy0 = 0, gx1 = 0, gy1 = 0; /* pixel box the launch covers */
    /* op masks per band of the frame (KSpill::masks): offset into frame_plan::masks, in uint64 */
    size_t   mask_off = 0;
    int      ncols = 0, nrows = 0, nchunks = 0, col_shift = 6;
    bool     dense = false; /* every op is in every band */
};

struct frame_plan {
    std::vector<KOp> ops;
    std::vector<segment> segments; /* never empty: an empty scene is one launch that stores the zero frame */
    uint32_t         spill_mask = 0; /* KSurf bits (RGBA1, RGBA2) that need a buffer in HBM between launches */
    int              n_key_bufs = 0; /* 1-channel buffers the launches hand key surfaces over through */
    bool             full = false; /* needs the RGBA1/RGBA2 register surfaces */
    bool             mix_pairs = false; /* holds fused MIX pairs (BB_MIX_HEAD): a frame of the fast class without temporaries */
    bool             fast = false; /* every draw qualifies for cghip_composite_fast (then there is one segment) */
    std::vector<cghip_plane*> planes; /* every plane referenced by an emitted op */
    std::vector<uint64_t>     masks;  /* the segments' op masks; uploaded behind the ops */
};

/* image_renderer / image_kernel host logic -> KOp list. Pure host code (no HIP calls). */
int plan_frame(const cghip_format* fmt, const cghip_surface* surfaces, int nsurfaces, const cghip_draw* draws,
               int ndraws, frame_plan* out);

} // namespace cghip

extern "C" hipError_t cghip_launch_composite(const KOp* d_ops, int nops, void* d_out, int W, int H, int bits,
                                             int field_mode, int full, int fast, const KSpill* sp, int gw, int gh,
                                             hipStream_t stream);

/* device address of this GPU's static scratch array (cghip_kernels.hip); looked up once per device, no allocation */
extern "C" hipError_t cghip_scratch_address(int gpu, void** ptr, size_t* bytes);

#endif
