/*
 * host_capi.cpp -- a flat C shim over the C++ host mixer (image_mixer.h) so that the Python test-suite and
 * bench.py can drive push / visit / pop / render exactly as core::mixer does (core/mixer/mixer.cpp:62-92).
 * Not part of the drop-in boundary (that is include/cghip.h); this only makes the C++ host side scriptable.
 */
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "image_mixer.h"

using namespace caspar;
namespace hip = caspar::accelerator::hip;

namespace {
thread_local std::string g_err;
hip::accelerator&        the_accelerator()
{
    static hip::accelerator a; /* shell/server.cpp:101 owns exactly one accelerator */
    return a;
}

struct host_mixer
{
    std::unique_ptr<hip::image_mixer> mixer;
    std::shared_ptr<hip::device>      dev;
};

/* Filling a pinned plane stands in for a decoder writing its output (modules/ffmpeg/util/av_util.cpp:40-80 copies
 * row by row inside tbb::parallel_for). One thread moves ~10 GB/s, a 2160p 16-layer frame is 0.5 GB: the copy is
 * split over a few persistent workers so that it keeps up with the PCIe upload that follows it. */
class copy_pool
{
    std::vector<std::thread>          workers_;
    std::mutex                        mu_;
    std::condition_variable           cv_, done_cv_;
    std::vector<std::function<void()>> tasks_;
    int                               pending_ = 0;
    bool                              stop_    = false;

  public:
    explicit copy_pool(int n)
    {
        for (int i = 0; i < n; ++i)
            workers_.emplace_back([this] {
                for (;;) {
                    std::function<void()> task;
                    {
                        std::unique_lock<std::mutex> l(mu_);
                        cv_.wait(l, [this] { return stop_ || !tasks_.empty(); });
                        if (stop_ && tasks_.empty())
                            return;
                        task = std::move(tasks_.back());
                        tasks_.pop_back();
                    }
                    task();
                    std::lock_guard<std::mutex> l(mu_);
                    if (--pending_ == 0)
                        done_cv_.notify_all();
                }
            });
    }
    ~copy_pool()
    {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& w : workers_)
            w.join();
    }
    void copy(void* dst, const void* src, std::size_t bytes)
    {
        const std::size_t chunk = std::size_t(2) << 20;
        if (bytes <= 2 * chunk || workers_.empty()) {
            std::memcpy(dst, src, bytes);
            return;
        }
        std::unique_lock<std::mutex> l(mu_);
        for (std::size_t off = 0; off < bytes; off += chunk) {
            const std::size_t n = std::min(chunk, bytes - off);
            tasks_.emplace_back([=] { std::memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, n); });
            ++pending_;
        }
        cv_.notify_all();
        done_cv_.wait(l, [this] { return pending_ == 0; });
    }
};

copy_pool& the_copy_pool()
{
    static copy_pool p([] {
        const char* e = std::getenv("CGHOST_COPY_THREADS");
        int         n = e ? std::atoi(e) : 8;
        return std::max(0, std::min(n, 32));
    }());
    return p;
}

void from_pod(const cghip_transform& o, core::image_transform& t)
{
    t.opacity = o.opacity, t.contrast = o.contrast, t.brightness = o.brightness, t.saturation = o.saturation;
    for (int i = 0; i < 2; ++i) {
        t.anchor[i] = o.anchor[i], t.fill_translation[i] = o.fill_translation[i], t.fill_scale[i] = o.fill_scale[i];
        t.clip_translation[i] = o.clip_translation[i], t.clip_scale[i] = o.clip_scale[i];
        t.crop.ul[i] = o.crop_ul[i], t.crop.lr[i] = o.crop_lr[i];
        t.perspective.ul[i] = o.pers_ul[i], t.perspective.ur[i] = o.pers_ur[i];
        t.perspective.lr[i] = o.pers_lr[i], t.perspective.ll[i] = o.pers_ll[i];
    }
    t.angle             = o.angle;
    t.levels.min_input  = o.levels_min_input;
    t.levels.max_input  = o.levels_max_input;
    t.levels.gamma      = o.levels_gamma;
    t.levels.min_output = o.levels_min_output;
    t.levels.max_output = o.levels_max_output;
    t.chroma.enable = o.chroma_enable != 0, t.chroma.show_mask = o.chroma_show_mask != 0;
    t.chroma.target_hue                = o.chroma_target_hue;
    t.chroma.hue_width                 = o.chroma_hue_width;
    t.chroma.min_saturation            = o.chroma_min_saturation;
    t.chroma.min_brightness            = o.chroma_min_brightness;
    t.chroma.softness                  = o.chroma_softness;
    t.chroma.spill_suppress            = o.chroma_spill_suppress;
    t.chroma.spill_suppress_saturation = o.chroma_spill_suppress_saturation;
    t.is_key = o.is_key != 0, t.invert = o.invert != 0, t.is_mix = o.is_mix != 0;
    t.blend_mode  = static_cast<core::blend_mode>(o.blend_mode);
    t.layer_depth = o.layer_depth;
}

template <class F>
int guarded(F&& f)
{
    try {
        f();
        return 0;
    } catch (const invalid_argument& e) {
        g_err = e.what();
        return CGHIP_ERR_INVALID;
    } catch (const not_supported& e) {
        g_err = e.what();
        return CGHIP_ERR_UNSUPPORTED;
    } catch (const std::exception& e) {
        g_err = e.what();
        return CGHIP_ERR_HIP;
    }
}
} // namespace

extern "C" {

struct cghost_plane
{
    int32_t     width, height, stride, bits;
    const void* data;
};

const char* cghost_last_error(void) { return g_err.c_str(); }

int cghost_device_count(void) { return the_accelerator().device_count(); }

/* gpu_index >= 0: that GPU; -1: accelerator's round-robin placement; -2: dry run (no device, flatten only);
 * <= -100: a device object of its own on GPU (-100 - gpu_index), not shared with the accelerator's per-GPU one: its
 * planes and pools are foreign to every other mixer, which is how a route between two GPUs looks to the host code
 * (lets the routed path be exercised on a one-GPU box). */
int cghost_mixer_create(int channel_id, int gpu_index, void** out)
{
    return guarded([&] {
        auto hm = std::make_unique<host_mixer>();
        if (gpu_index == -2) {
            hm->mixer = std::make_unique<hip::image_mixer>(nullptr, channel_id);
        } else if (gpu_index <= -100) {
            hm->dev   = std::make_shared<hip::device>(-100 - gpu_index);
            hm->mixer = std::make_unique<hip::image_mixer>(hm->dev, channel_id);
        } else if (gpu_index == -1) {
            hm->mixer = the_accelerator().create_image_mixer(channel_id);
        } else {
            hm->mixer = the_accelerator().create_image_mixer(channel_id, gpu_index);
            hm->dev   = the_accelerator().get_device(gpu_index);
        }
        *out = hm.release();
    });
}

/* Another channel on the device object `other` renders on (the accelerator's shared one or a private stand-in): how N
 * channels share one stand-in "GPU" in tests/native/route_driver.cpp. */
int cghost_mixer_create_on(void* other, int channel_id, void** out)
{
    return guarded([&] {
        auto* o = static_cast<host_mixer*>(other);
        if (!o || !o->dev)
            throw invalid_argument("mixer has no device");
        auto hm   = std::make_unique<host_mixer>();
        hm->dev   = o->dev;
        hm->mixer = std::make_unique<hip::image_mixer>(hm->dev, channel_id);
        *out      = hm.release();
    });
}

int cghost_mixer_destroy(void* h)
{
    delete static_cast<host_mixer*>(h);
    return 0;
}

int cghost_mixer_push(void* h, const cghip_transform* t)
{
    return guarded([&] {
        core::frame_transform ft;
        from_pod(*t, ft.image_transform);
        static_cast<host_mixer*>(h)->mixer->push(ft);
    });
}

/* via_factory != 0: create_frame -> fill pinned planes -> const_frame (commit uploads), the producer path
 * (modules/ffmpeg/util/av_util.cpp:40-80). Otherwise a foreign const_frame over the caller's memory. */
int cghost_mixer_visit(void* h, int pixel_format, int nplanes, const cghost_plane* planes, const cghip_coord* coords,
                       int ncoords, int via_factory)
{
    return guarded([&] {
        auto&                   mixer = *static_cast<host_mixer*>(h)->mixer;
        core::pixel_format_desc desc(static_cast<core::pixel_format>(pixel_format));
        for (int n = 0; n < nplanes; ++n)
            desc.planes.emplace_back(planes[n].width, planes[n].height, planes[n].stride, planes[n].bits);
        std::vector<core::frame_geometry::coord> g;
        for (int i = 0; i < ncoords; ++i) {
            core::frame_geometry::coord c(coords[i].vertex_x, coords[i].vertex_y, coords[i].texture_x, coords[i].texture_y);
            c.texture_r = coords[i].texture_r;
            c.texture_q = coords[i].texture_q;
            g.push_back(c);
        }
        core::frame_geometry geometry(core::frame_geometry::geometry_type::quad, g);
        if (via_factory) {
            auto frame = mixer.create_frame(nullptr, desc);
            for (int n = 0; n < nplanes; ++n)
                the_copy_pool().copy(frame.image_data(static_cast<std::size_t>(n)).data(), planes[n].data,
                                     static_cast<std::size_t>(desc.planes[static_cast<std::size_t>(n)].size));
            frame.geometry() = geometry;
            mixer.visit(core::const_frame(std::move(frame)));
        } else {
            std::vector<array<const std::uint8_t>> data;
            for (int n = 0; n < nplanes; ++n)
                data.emplace_back(static_cast<const std::uint8_t*>(planes[n].data),
                                  static_cast<std::size_t>(desc.planes[static_cast<std::size_t>(n)].size), nullptr);
            core::const_frame frame(std::move(data), desc);
            frame.geometry() = geometry;
            mixer.visit(frame);
        }
    });
}

/* A still frame that stays resident in HBM and is re-visited every tick without a new upload, the way a paused
 * clip or an image producer's frame is (modules/image/producer/image_producer.cpp:89-100 creates it once). */
int cghost_frame_create(void* h, int pixel_format, int nplanes, const cghost_plane* planes, const cghip_coord* coords,
                        int ncoords, void** out)
{
    return guarded([&] {
        auto&                   mixer = *static_cast<host_mixer*>(h)->mixer;
        core::pixel_format_desc desc(static_cast<core::pixel_format>(pixel_format));
        for (int n = 0; n < nplanes; ++n)
            desc.planes.emplace_back(planes[n].width, planes[n].height, planes[n].stride, planes[n].bits);
        auto frame = mixer.create_frame(nullptr, desc);
        for (int n = 0; n < nplanes; ++n)
            std::memcpy(frame.image_data(static_cast<std::size_t>(n)).data(), planes[n].data,
                        static_cast<std::size_t>(desc.planes[static_cast<std::size_t>(n)].size));
        if (ncoords == 4) {
            std::vector<core::frame_geometry::coord> g;
            for (int i = 0; i < 4; ++i) {
                core::frame_geometry::coord c(coords[i].vertex_x, coords[i].vertex_y, coords[i].texture_x, coords[i].texture_y);
                c.texture_r = coords[i].texture_r;
                c.texture_q = coords[i].texture_q;
                g.push_back(c);
            }
            frame.geometry() = core::frame_geometry(core::frame_geometry::geometry_type::quad, g);
        }
        *out = new core::const_frame(std::move(frame));
    });
}

int cghost_frame_destroy(void* frame)
{
    delete static_cast<core::const_frame*>(frame);
    return 0;
}

int cghost_mixer_visit_frame(void* h, void* frame)
{
    return guarded([&] { static_cast<host_mixer*>(h)->mixer->visit(*static_cast<core::const_frame*>(frame)); });
}

int cghost_mixer_pop(void* h)
{
    return guarded([&] { static_cast<host_mixer*>(h)->mixer->pop(); });
}

static core::video_format_desc to_desc(const cghip_format* f)
{
    core::video_format_desc d(f->width, f->height, f->square_width, f->square_height, 1, 50.0, L"custom");
    d.bits       = f->bits;
    d.field_mode = f->field_mode;
    return d;
}

/* render + get(): copies the frame into out (if non-null). readback = 0 keeps the frame in HBM. */
int cghost_mixer_render(void* h, const cghip_format* f, void* out, size_t capacity, int readback, float* kernel_ms)
{
    return guarded([&] {
        auto& mixer = *static_cast<host_mixer*>(h)->mixer;
        mixer.set_readback(readback != 0);
        mixer.set_field_rows_only(readback == 2);
        auto fut   = mixer(to_desc(f));
        auto frame = fut.get();
        if (out && readback) {
            if (frame.size() > capacity)
                throw invalid_argument("output buffer too small");
            std::memcpy(out, frame.data(), frame.size());
        }
        if (kernel_ms)
            *kernel_ms = mixer.last_kernel_ms();
    });
}

/* the pending scene composited into the caller's device memory by the entry point that allocates nothing */
int cghost_mixer_render_noalloc(void* h, const cghip_format* f, void* device_ptr, size_t bytes)
{
    return guarded([&] { static_cast<host_mixer*>(h)->mixer->render_into(to_desc(f), device_ptr, bytes); });
}

/* operator() without the get(): the deferred future is returned as a handle (core/mixer/mixer.cpp:70-91 keeps it
 * in a queue and calls get() one tick later). */
struct pending_frame
{
    std::future<array<const std::uint8_t>> fut;
    hip::image_mixer*                      mixer;
    bool                                   readback;
};

int cghost_mixer_set_kernel_timing(void* h, int on)
{
    return guarded([&] { static_cast<host_mixer*>(h)->mixer->set_kernel_timing(on != 0); });
}

int cghost_mixer_render_async(void* h, const cghip_format* f, int readback, void** out)
{
    return guarded([&] {
        auto& mixer = *static_cast<host_mixer*>(h)->mixer;
        mixer.set_readback(readback != 0);
        mixer.set_field_rows_only(readback == 2); /* 2: only the rows of format.field_mode's field cross PCIe */
        *out = new pending_frame{mixer(to_desc(f)), &mixer, readback != 0};
    });
}

int cghost_future_get(void* pending, void* out, size_t capacity, float* kernel_ms)
{
    std::unique_ptr<pending_frame> p(static_cast<pending_frame*>(pending));
    return guarded([&] {
        auto frame = p->fut.get();
        if (out && p->readback) {
            if (frame.size() > capacity)
                throw invalid_argument("output buffer too small");
            std::memcpy(out, frame.data(), frame.size());
        }
        if (kernel_ms)
            *kernel_ms = p->mixer->last_kernel_ms();
    });
}

/* host-logic introspection: the draw list the renderer would issue for the pending scene */
int cghost_mixer_flatten(void* h, const cghip_format* f, cghip_draw* draws, int capacity, int* ndraws,
                         cghip_surface* surfaces, int* nsurfaces)
{
    return guarded([&] {
        std::vector<cghip_surface> surf;
        auto                       calls = static_cast<host_mixer*>(h)->mixer->flatten(to_desc(f), &surf);
        *ndraws                          = static_cast<int>(calls.size());
        for (int i = 0; i < *ndraws && i < capacity; ++i)
            draws[i] = calls[static_cast<std::size_t>(i)].draw;
        if (nsurfaces)
            *nsurfaces = static_cast<int>(surf.size());
        if (surfaces)
            for (std::size_t i = 0; i < surf.size() && i < CGHIP_MAX_SURFACES; ++i)
                surfaces[i] = surf[i];
    });
}

int cghost_device_info(int gpu_index, char* json, size_t capacity)
{
    return guarded([&] {
        auto s = the_accelerator().get_device(gpu_index)->info();
        if (s.size() + 1 > capacity)
            throw invalid_argument("info buffer too small");
        std::memcpy(json, s.c_str(), s.size() + 1);
    });
}

/* info of the device object a mixer renders on (the accelerator's shared one or a private one) */
int cghost_mixer_device_info(void* h, char* json, size_t capacity)
{
    return guarded([&] {
        auto* hm = static_cast<host_mixer*>(h);
        if (!hm || !hm->dev)
            throw invalid_argument("mixer has no device");
        auto s = hm->dev->info();
        if (s.size() + 1 > capacity)
            throw invalid_argument("info buffer too small");
        std::memcpy(json, s.c_str(), s.size() + 1);
    });
}

int cghost_device_gc(int gpu_index)
{
    return guarded([&] { the_accelerator().get_device(gpu_index)->gc().get(); });
}

} /* extern "C" */
