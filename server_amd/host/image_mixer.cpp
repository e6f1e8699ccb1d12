/*
 * accelerator/hip image mixer -- host side above the C ABI. See image_mixer.h.
 *
 * What it keeps from the reference (accelerator/ogl/image/image_mixer.cpp):
 *   :54-72    item / layer tree
 *   :247-299  push / visit / pop (transform stack, a new layer whenever layer_depth rises)
 *   :104-225  image_renderer: the ORDER of kernel draws incl. key / mix / blend-mode temporaries
 *   :306-331  create_frame: one pinned, host-writable buffer per plane; upload starts at commit
 * What changes: the draws are not executed one by one; they are flattened into a cghip_draw list and
 * handed to cghip_mixer_render_async, which fuses them into one kernel launch. Temporaries become
 * surface slot numbers.
 *
 * Provenance: this file is the adapter between CasparCG Server's plugin interface and the C ABI, and three pieces of
 * it restate the reference's own logic closely, on purpose, identifier names included, so that the adapter behaves like
 * the code it replaces: image_transform::operator*= (core/frame/frame_transform.cpp:53-108), the item / layer structs
 * (accelerator/ogl/image/image_mixer.cpp:54-72) and push / visit / pop (:247-299). CasparCG Server is Copyright (c)
 * Sveriges Television AB and licensed under the GNU General Public License v3 or later; those restated parts, and this
 * adapter as a derived work meant to be compiled into the reference tree (integration/), fall under the same licence.
 * Everything below the C ABI (server_amd/csrc) has no counterpart in the reference.
 */
#include "image_mixer.h"

extern "C" int cghip_debug_plane_stub(int width, int height, int channels, int bits, cghip_plane** out); /* cghip_debug.h */

#include <algorithm>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>

namespace caspar {

namespace core {

/* core/frame/frame_transform.cpp:53-108 */
image_transform& image_transform::operator*=(const image_transform& other)
{
    opacity *= other.opacity;
    brightness *= other.brightness;
    contrast *= other.contrast;
    saturation *= other.saturation;

    const double aspect_ratio = 1.0; /* fixed in the reference (:61) */
    const double ox = other.fill_translation[0], oy = other.fill_translation[1] / aspect_ratio;
    const double rx = ox * std::cos(angle) - oy * std::sin(angle);
    const double ry = (ox * std::sin(angle) + oy * std::cos(angle)) * aspect_ratio;

    for (int i = 0; i < 2; ++i)
        anchor[i] += other.anchor[i] * fill_scale[i];
    fill_translation[0] += rx * fill_scale[0];
    fill_translation[1] += ry * fill_scale[1];
    for (int i = 0; i < 2; ++i) {
        fill_scale[i] *= other.fill_scale[i];
        clip_translation[i] += other.clip_translation[i] * clip_scale[i];
        clip_scale[i] *= other.clip_scale[i];
    }
    angle += other.angle;

    for (int i = 0; i < 2; ++i) { /* transform_rect / transform_corners :32-51 */
        crop.ul[i] += other.crop.ul[i];
        crop.lr[i] *= other.crop.lr[i];
        perspective.ul[i] += other.perspective.ul[i];
        perspective.lr[i] *= other.perspective.lr[i];
    }
    perspective.ur[0] *= other.perspective.ur[0];
    perspective.ur[1] += other.perspective.ur[1];
    perspective.ll[0] += other.perspective.ll[0];
    perspective.ll[1] *= other.perspective.ll[1];

    levels.min_input  = std::max(levels.min_input, other.levels.min_input);
    levels.max_input  = std::min(levels.max_input, other.levels.max_input);
    levels.min_output = std::max(levels.min_output, other.levels.min_output);
    levels.max_output = std::min(levels.max_output, other.levels.max_output);
    levels.gamma *= other.levels.gamma;
    chroma.enable |= other.chroma.enable;
    chroma.show_mask |= other.chroma.show_mask;
    chroma.target_hue     = std::max(other.chroma.target_hue, chroma.target_hue);
    chroma.min_saturation = std::max(other.chroma.min_saturation, chroma.min_saturation);
    chroma.min_brightness = std::max(other.chroma.min_brightness, chroma.min_brightness);
    chroma.hue_width      = std::max(other.chroma.hue_width, chroma.hue_width);
    chroma.softness       = std::max(other.chroma.softness, chroma.softness);
    chroma.spill_suppress = std::max(other.chroma.spill_suppress, chroma.spill_suppress);
    chroma.spill_suppress_saturation = std::min(other.chroma.spill_suppress_saturation, chroma.spill_suppress_saturation);
    is_key |= other.is_key;
    invert |= other.invert;
    is_mix |= other.is_mix;
    blend_mode = std::max(blend_mode, other.blend_mode);
    layer_depth += other.layer_depth;
    return *this;
}

} // namespace core

namespace accelerator { namespace hip {

namespace {

[[noreturn]] void throw_status(int rc, const char* what)
{
    std::string msg = std::string(what) + ": " + cghip_last_error();
    if (rc == CGHIP_ERR_INVALID)
        throw invalid_argument(msg);
    if (rc == CGHIP_ERR_NO_DEVICE || rc == CGHIP_ERR_UNSUPPORTED)
        throw not_supported(msg);
    throw caspar_exception(msg);
}
void check(int rc, const char* what)
{
    if (rc != CGHIP_OK)
        throw_status(rc, what);
}

using plane_ptr = std::shared_ptr<cghip_plane>;

plane_ptr own(cghip_plane* p)
{
    return plane_ptr(p, [](cghip_plane* q) { cghip_plane_release(q); });
}

/* what const_frame::opaque() carries for frames created by this mixer: uploaded planes + the pinned
 * arrays they are being copied from (kept until the frame that uses them has rendered).
 *
 * Routes (core/producer/route/route_producer.cpp:40-126): a frame made on one channel is visited by
 * mixers of other channels, possibly on another GPU. `home` names the device the planes live on; a mixer on another
 * device gets a device-to-device copy over xGMI (cghip_plane_peer_copy_async), and the copy is cached HERE, per
 * destination device, so that N routes of one frame onto a GPU (or N ticks showing the same still) copy it once. The
 * cache dies with the frame. */
struct uploaded;
/* The opaques this accelerator made and that are still alive. const_frame::opaque() is type-erased (boost::any in the
 * reference, where the pointer form of any_cast answers the question); here a foreign opaque must be told from ours
 * WITHOUT touching the pointee, so visit() looks the pointer up. */
struct uploaded_registry
{
    std::mutex                mu;
    std::set<const uploaded*> live;
    static uploaded_registry& get()
    {
        static uploaded_registry r;
        return r;
    }
    void add(const uploaded* u)
    {
        std::lock_guard<std::mutex> g(mu);
        live.insert(u);
    }
    void remove(const uploaded* u)
    {
        std::lock_guard<std::mutex> g(mu);
        live.erase(u);
    }
    bool ours(const void* p)
    {
        std::lock_guard<std::mutex> g(mu);
        return live.count(static_cast<const uploaded*>(p)) != 0;
    }
};

struct uploaded
{
    /* declared first = destroyed last: planes return their memory to their device's pools on release, and the pinned
       buffers an upload is still reading (`sources`) outlive the planes, whose release waits for that upload */
    std::shared_ptr<device>                home_keep;   /* the device the planes live on */
    /* routed copy: what of the frame it was copied from must outlive the copy - its device and the pinned buffers its
       upload may still be reading (the C ABI itself retains the source PLANES until the copy is released). Handles
       only, never the origin `uploaded` itself: the origin's `routed` cache holds this copy, and a strong pointer back
       would keep both alive for ever (planes, pinned buffers and registry entries of every routed frame leaked). */
    std::shared_ptr<device>                origin_home_keep;
    std::vector<array<const std::uint8_t>> origin_sources;
    std::vector<array<const std::uint8_t>> sources;
    std::vector<plane_ptr>                 planes;
    cghip_device*                          home  = nullptr;
    uploaded() { uploaded_registry::get().add(this); }
    ~uploaded() { uploaded_registry::get().remove(this); }
    uploaded(const uploaded&) = delete;
    std::mutex                                                        route_mu;
    std::vector<std::pair<cghip_device*, std::shared_ptr<uploaded>>> routed;

    std::shared_ptr<uploaded> on_device(const std::shared_ptr<device>& dst_dev)
    {
        cghip_device*               dst = dst_dev->handle();
        std::lock_guard<std::mutex> g(route_mu);
        for (auto& r : routed)
            if (r.first == dst)
                return r.second;
        auto copy       = std::make_shared<uploaded>();
        copy->home      = dst;
        copy->home_keep        = dst_dev;
        copy->origin_home_keep = home_keep;
        copy->origin_sources   = sources;
        for (auto& pl : planes) {
            cghip_plane* out = nullptr; /* the copy retains its source plane (cghip.h) */
            check(cghip_plane_peer_copy_async(dst, pl.get(), &out), "cghip_plane_peer_copy_async");
            copy->planes.push_back(own(out));
        }
        routed.emplace_back(dst, copy);
        return copy;
    }
};

struct item /* image_mixer.cpp:54-60 */
{
    core::pixel_format_desc   pix_desc = core::pixel_format::invalid;
    std::shared_ptr<uploaded> textures;
    core::image_transform     transform;
    core::frame_geometry      geometry = core::frame_geometry::get_default();
};

struct layer /* image_mixer.cpp:62-72 */
{
    std::vector<layer> sublayers;
    std::vector<item>  items;
    core::blend_mode   blend_mode;
    explicit layer(core::blend_mode blend_mode)
        : blend_mode(blend_mode)
    {
    }
};

void to_pod(const core::image_transform& t, cghip_transform& o)
{
    o = cghip_transform{};
    o.opacity = t.opacity, o.contrast = t.contrast, o.brightness = t.brightness, o.saturation = t.saturation;
    for (int i = 0; i < 2; ++i) {
        o.anchor[i] = t.anchor[i], o.fill_translation[i] = t.fill_translation[i], o.fill_scale[i] = t.fill_scale[i];
        o.clip_translation[i] = t.clip_translation[i], o.clip_scale[i] = t.clip_scale[i];
        o.crop_ul[i] = t.crop.ul[i], o.crop_lr[i] = t.crop.lr[i];
        o.pers_ul[i] = t.perspective.ul[i], o.pers_ur[i] = t.perspective.ur[i];
        o.pers_lr[i] = t.perspective.lr[i], o.pers_ll[i] = t.perspective.ll[i];
    }
    o.angle             = t.angle;
    o.levels_min_input  = t.levels.min_input;
    o.levels_max_input  = t.levels.max_input;
    o.levels_gamma      = t.levels.gamma;
    o.levels_min_output = t.levels.min_output;
    o.levels_max_output = t.levels.max_output;
    o.chroma_enable = t.chroma.enable, o.chroma_show_mask = t.chroma.show_mask;
    o.chroma_target_hue                = t.chroma.target_hue;
    o.chroma_hue_width                 = t.chroma.hue_width;
    o.chroma_min_saturation            = t.chroma.min_saturation;
    o.chroma_min_brightness            = t.chroma.min_brightness;
    o.chroma_softness                  = t.chroma.softness;
    o.chroma_spill_suppress            = t.chroma.spill_suppress;
    o.chroma_spill_suppress_saturation = t.chroma.spill_suppress_saturation;
    o.is_key = t.is_key, o.invert = t.invert, o.is_mix = t.is_mix;
    o.blend_mode  = static_cast<int>(t.blend_mode);
    o.layer_depth = t.layer_depth;
}

/* image_renderer (image_mixer.cpp:74-226) turned into a recorder: same control flow, but every
 * kernel_.draw becomes a draw_call and every create_texture a new surface slot. */
class draw_recorder
{
  public:
    draw_recorder(const core::video_format_desc& format_desc, std::vector<draw_call>& out,
                  std::vector<cghip_surface>& surfaces)
        : format_desc_(format_desc)
        , out_(out)
        , surfaces_(surfaces)
    {
        surfaces_.clear();
        surfaces_.push_back(cghip_surface{4, 0}); /* target_texture :95 */
    }

    static constexpr int none = -1;

    void draw(int target, const std::vector<layer>& layers) /* :104-114 */
    {
        int layer_key = none;
        for (auto& l : layers) {
            draw(target, l.sublayers);
            draw(target, l, layer_key);
        }
    }

  private:
    int create_surface(int channels)
    {
        if (surfaces_.size() >= CGHIP_MAX_SURFACES)
            throw not_supported("scene needs more than CGHIP_MAX_SURFACES render temporaries");
        surfaces_.push_back(cghip_surface{channels, 0});
        return static_cast<int>(surfaces_.size()) - 1;
    }

    void draw(int target, const layer& l, int& layer_key) /* :116-154 */
    {
        if (l.items.empty())
            return;
        int local_key = none, local_mix = none;
        if (l.blend_mode != core::blend_mode::normal) {
            int layer_texture = create_surface(4);
            for (auto& it : l.items)
                draw(layer_texture, it, layer_key, local_key, local_mix);
            draw(layer_texture, local_mix, core::blend_mode::normal);
            draw(target, layer_texture, l.blend_mode);
        } else {
            for (auto& it : l.items)
                draw(target, it, layer_key, local_key, local_mix);
            draw(target, local_mix, core::blend_mode::normal);
        }
        layer_key = local_key;
    }

    void draw(int target, const item& it, int& layer_key, int& local_key, int& local_mix) /* :156-205 */
    {
        draw_call dc{};
        cghip_draw& d    = dc.draw;
        d.pixel_format   = static_cast<int>(it.pix_desc.format);
        d.nplanes        = static_cast<int>(it.textures ? it.textures->planes.size() : 0);
        d.source_surface = none;
        for (int n = 0; n < d.nplanes && n < 4; ++n) {
            d.planes[n] = it.textures->planes[n].get();
            dc.keep.push_back(it.textures->planes[n]);
        }
        to_pod(it.transform, d.transform);
        d.geometry_is_quad = it.geometry.type() == core::frame_geometry::geometry_type::quad;
        d.ncoords          = static_cast<int>(std::min<std::size_t>(it.geometry.data().size(), 4));
        for (int i = 0; i < d.ncoords; ++i) {
            auto& c     = it.geometry.data()[i];
            d.coords[i] = cghip_coord{c.vertex_x, c.vertex_y, c.texture_x, c.texture_y, c.texture_r, c.texture_q};
        }
        d.blend_mode   = static_cast<int>(core::blend_mode::normal);
        d.keyer        = CGHIP_KEYER_LINEAR;
        d.aspect_ratio = static_cast<double>(format_desc_.square_width) / static_cast<double>(format_desc_.square_height);

        if (it.transform.is_key) {
            if (local_key == none)
                local_key = create_surface(1);
            d.target = local_key, d.local_key = none, d.layer_key = none;
            out_.push_back(std::move(dc));
        } else if (it.transform.is_mix) {
            if (local_mix == none)
                local_mix = create_surface(4);
            d.target    = local_mix;
            d.local_key = local_key;
            local_key   = none; /* std::move(local_key_texture) */
            d.layer_key = layer_key;
            d.keyer     = CGHIP_KEYER_ADDITIVE;
            out_.push_back(std::move(dc));
        } else {
            draw(target, local_mix, core::blend_mode::normal);
            d.target    = target;
            d.local_key = local_key;
            local_key   = none;
            d.layer_key = layer_key;
            out_.push_back(std::move(dc));
        }
    }

    /* :207-225. `source_buffer` is a named rvalue reference in the reference and spl::make_shared_ptr takes its
       argument by value (common/memory.h:633): the caller's texture is copied, never moved from. A layer's mix texture
       so stays set: it is composited again before every later plain item (:197) and at the layer's end (:138/:150). */
    void draw(int target, const int& source_buffer, core::blend_mode blend_mode)
    {
        if (source_buffer == none)
            return;
        draw_call dc{};
        cghip_draw& d    = dc.draw;
        d.pixel_format   = CGHIP_PF_BGRA;
        d.nplanes        = 0;
        d.source_surface = source_buffer;
        d.target         = target;
        d.local_key = d.layer_key = none;
        d.blend_mode     = static_cast<int>(blend_mode);
        d.keyer          = CGHIP_KEYER_LINEAR;
        to_pod(core::image_transform(), d.transform);
        d.geometry_is_quad = 1;
        d.ncoords          = 4;
        auto& g            = core::frame_geometry::get_default().data();
        for (int i = 0; i < 4; ++i)
            d.coords[i] = cghip_coord{g[i].vertex_x, g[i].vertex_y, g[i].texture_x, g[i].texture_y, g[i].texture_r, g[i].texture_q};
        d.aspect_ratio = 1.0;
        out_.push_back(std::move(dc));
    }

    const core::video_format_desc& format_desc_;
    std::vector<draw_call>&        out_;
    std::vector<cghip_surface>&    surfaces_;
};

} // namespace

/* ------------------------------------------------------------------ device */

device::device(int gpu_index)
    : gpu_(gpu_index)
{
    check(cghip_device_create(gpu_index, &dev_), "cghip_device_create");
}
device::~device() { cghip_device_destroy(dev_); }

std::string device::info() const
{
    char buf[2048];
    check(cghip_device_info(dev_, buf, sizeof buf), "cghip_device_info");
    return buf;
}

std::future<void> device::gc()
{
    return std::async(std::launch::deferred, [this] { check(cghip_device_gc(dev_), "cghip_device_gc"); });
}

/* ------------------------------------------------------------------ image_mixer */

struct image_mixer::impl : std::enable_shared_from_this<image_mixer::impl>
{
    std::shared_ptr<device>            dev_;
    cghip_mixer*                       mixer_ = nullptr;
    std::vector<core::image_transform> transform_stack_;
    std::vector<layer>                 layers_;
    std::vector<int>                   layer_path_; /* layer_stack_ (:236) as an index path: vectors may reallocate */
    float                              last_ms_  = 0.0f;
    bool                               readback_ = true;
    bool                               field_rows_only_ = false;
    bool                               kernel_timing_   = true;

    impl(const std::shared_ptr<device>& dev, int channel_id)
        : dev_(dev)
        , transform_stack_(1)
    {
        /* dev == nullptr: dry run for host-logic tests -- scenes can be built and flattened, not rendered */
        if (dev_)
            check(cghip_mixer_create(dev_->handle(), channel_id, &mixer_), "cghip_mixer_create");
    }
    ~impl()
    {
        if (mixer_)
            cghip_mixer_destroy(mixer_);
    }

    layer* current_layer()
    {
        if (layer_path_.empty())
            return nullptr;
        layer* l = &layers_[static_cast<std::size_t>(layer_path_[0])];
        for (std::size_t d = 1; d < layer_path_.size(); ++d)
            l = &l->sublayers[static_cast<std::size_t>(layer_path_[d])];
        return l;
    }

    void push(const core::frame_transform& transform) /* :247-264 */
    {
        auto previous_layer_depth = transform_stack_.back().layer_depth;
        transform_stack_.push_back(transform_stack_.back() * transform.image_transform);
        auto new_layer_depth = transform_stack_.back().layer_depth;

        if (previous_layer_depth < new_layer_depth) {
            layer new_layer(transform_stack_.back().blend_mode);
            if (layer_path_.empty()) {
                layers_.push_back(std::move(new_layer));
                layer_path_.push_back(static_cast<int>(layers_.size()) - 1);
            } else {
                auto* cur = current_layer();
                cur->sublayers.push_back(std::move(new_layer));
                layer_path_.push_back(static_cast<int>(cur->sublayers.size()) - 1);
            }
        }
    }

    void visit(const core::const_frame& frame) /* :266-293 */
    {
        if (frame.pixel_format_desc().format == core::pixel_format::invalid)
            return;
        if (frame.pixel_format_desc().planes.empty())
            return;
        auto* cur = current_layer();
        if (!cur)
            throw invalid_argument("visit() outside a layer: no transform with layer_depth > 0 was pushed");

        item it;
        it.pix_desc  = frame.pixel_format_desc();
        it.transform = transform_stack_.back();
        it.geometry  = frame.geometry();

        /* pointer-form any_cast: tolerate empty / foreign opaque (SURVEY 8b ownership) */
        std::shared_ptr<uploaded> up;
        if (frame.opaque() && uploaded_registry::get().ours(frame.opaque().get()))
            up = std::static_pointer_cast<uploaded>(frame.opaque());
        if (up && up->planes.size() == it.pix_desc.planes.size()) {
            /* a routed frame whose planes live on another device is copied over once per destination device */
            it.textures = (dev_ && up->home && up->home != dev_->handle()) ? up->on_device(dev_) : up;
        } else {
            it.textures = upload(it.pix_desc, [&](int n) { return frame.image_data(static_cast<std::size_t>(n)); });
        }
        cur->items.push_back(std::move(it));
    }

    void pop() /* :295-299 */
    {
        transform_stack_.pop_back();
        layer_path_.resize(static_cast<std::size_t>(std::max(0, transform_stack_.back().layer_depth)));
    }

    template <class F>
    std::shared_ptr<uploaded> upload(const core::pixel_format_desc& desc, F&& source_of)
    {
        auto up  = std::make_shared<uploaded>();
        up->home      = dev_ ? dev_->handle() : nullptr;
        up->home_keep = dev_;
        for (int n = 0; n < static_cast<int>(desc.planes.size()); ++n) {
            auto&        pl  = desc.planes[static_cast<std::size_t>(n)];
            auto         src = source_of(n);
            cghip_plane* p   = nullptr;
            /* uyvy is two views of the same bytes (RG8 w x h, RGBA8 w/2 x h: av_util.cpp:163-171); the reference
               uploads them twice (TODO at av_util.cpp:45). Upload once, alias the second. */
            if (dev_ && n == 1 && desc.format == core::pixel_format::uyvy && desc.planes.size() == 2 &&
                desc.planes[0].size == pl.size && desc.planes[0].bits == pl.bits) {
                check(cghip_plane_view(up->planes[0].get(), pl.width, pl.height, pl.stride, &p), "cghip_plane_view");
                up->planes.push_back(own(p));
                up->sources.push_back(src);
                continue;
            }
            if (dev_)
                check(cghip_plane_upload_async(dev_->handle(), src.data(), pl.width, pl.height, pl.stride, pl.bits, &p),
                      "cghip_plane_upload_async");
            else
                check(cghip_debug_plane_stub(pl.width, pl.height, pl.stride, pl.bits, &p), "cghip_debug_plane_stub");
            up->planes.push_back(own(p));
            up->sources.push_back(src);
        }
        return up;
    }

    std::vector<draw_call> flatten(const core::video_format_desc& format_desc, std::vector<cghip_surface>* surfaces)
    {
        std::vector<draw_call>     calls;
        std::vector<cghip_surface> local;
        draw_recorder              rec(format_desc, calls, surfaces ? *surfaces : local);
        rec.draw(0, layers_);
        return calls;
    }

    std::future<array<const std::uint8_t>> render(const core::video_format_desc& format_desc) /* :301-304, :86-101 */
    {
        auto layers = std::move(layers_);
        layers_.clear();
        layer_path_.clear();

        if (!mixer_ && !layers.empty())
            throw not_supported("this image_mixer was created without a device (dry run): it cannot render");
        if (layers.empty()) { /* :89-92 bypass the GPU with an empty frame */
            const std::size_t bytes = static_cast<std::size_t>(format_desc.width) * format_desc.height * 4 * (format_desc.bits / 8);
            auto              zero  = std::make_shared<std::vector<std::uint8_t>>(bytes, 0);
            std::promise<array<const std::uint8_t>> p;
            p.set_value(array<const std::uint8_t>(zero->data(), bytes, zero));
            return p.get_future();
        }

        std::vector<draw_call>     calls;
        std::vector<cghip_surface> surfaces;
        {
            draw_recorder rec(format_desc, calls, surfaces);
            rec.draw(0, layers);
        }
        std::vector<cghip_draw> draws;
        draws.reserve(calls.size());
        for (auto& c : calls)
            draws.push_back(c.draw);

        cghip_format fmt{format_desc.width, format_desc.height, format_desc.square_width, format_desc.square_height,
                         format_desc.bits,  format_desc.field_mode};
        cghip_ticket* ticket = nullptr;
        check(cghip_mixer_render_async(mixer_, &fmt, surfaces.data(), static_cast<int>(surfaces.size()), draws.data(),
                                       static_cast<int>(draws.size()),
                                       (readback_ ? CGHIP_RENDER_READBACK : 0u) |
                                           (field_rows_only_ ? CGHIP_RENDER_FIELD_ROWS_ONLY : 0u) |
                                           (kernel_timing_ ? 0u : CGHIP_RENDER_UNTIMED),
                                       &ticket),
              "cghip_mixer_render_async");
        /* consumers may hold the frame after the channel is gone (decklink / ndi queues): the buffer keeps its device */
        std::shared_ptr<cghip_ticket> owner(ticket, [dev = dev_](cghip_ticket* t) { cghip_ticket_release(t); });
        auto                          self     = shared_from_this();
        auto                          keep     = std::make_shared<std::vector<layer>>(std::move(layers));
        const bool                    readback = readback_;

        /* deferred like common/future.h:9-13: the channel thread calls get() one tick later (mixer.cpp:85-91) */
        return std::async(std::launch::deferred, [owner, self, keep, readback]() -> array<const std::uint8_t> {
            const std::uint8_t* data = nullptr;
            std::size_t         size = 0;
            check(cghip_ticket_wait(owner.get(), readback ? &data : nullptr, &size), "cghip_ticket_wait");
            float ms = 0.0f;
            if (cghip_ticket_kernel_ms(owner.get(), &ms, nullptr) == CGHIP_OK && ms >= 0.0f)
                self->last_ms_ = ms; /* an untimed frame (-1) leaves the last measured value in place */
            if (!readback) { /* device-resident frame: hand out the device pointer, owner keeps it alive */
                const void* dptr = nullptr;
                check(cghip_ticket_device_ptr(owner.get(), &dptr, &size), "cghip_ticket_device_ptr");
                return array<const std::uint8_t>(static_cast<const std::uint8_t*>(dptr), size, owner);
            }
            return array<const std::uint8_t>(data, size, owner); /* buffer returns to the pool on release */
        });
    }

    void render_into(const core::video_format_desc& format_desc, void* device_ptr, std::size_t bytes)
    {
        auto layers = std::move(layers_);
        layers_.clear();
        layer_path_.clear();
        if (!mixer_)
            throw not_supported("this image_mixer was created without a device (dry run): it cannot render");
        std::vector<draw_call>     calls;
        std::vector<cghip_surface> surfaces;
        {
            draw_recorder rec(format_desc, calls, surfaces);
            rec.draw(0, layers);
        }
        std::vector<cghip_draw> draws;
        draws.reserve(calls.size());
        for (auto& c : calls)
            draws.push_back(c.draw);
        cghip_format fmt{format_desc.width, format_desc.height, format_desc.square_width, format_desc.square_height,
                         format_desc.bits,  format_desc.field_mode};
        /* an empty scene is the zero frame here too: one launch with no ops stores it */
        check(cghip_mixer_render_noalloc(mixer_, &fmt, surfaces.data(), static_cast<int>(surfaces.size()), draws.data(),
                                         static_cast<int>(draws.size()), device_ptr, bytes),
              "cghip_mixer_render_noalloc");
    }

    core::mutable_frame create_frame(const void* tag, const core::pixel_format_desc& desc) /* :306-331 */
    {
        if (!dev_)
            throw not_supported("this image_mixer was created without a device (dry run): no pinned buffers");
        std::vector<array<std::uint8_t>> image_data;
        for (auto& plane : desc.planes) {
            void*          ptr = nullptr;
            cghip_hostbuf* hb  = nullptr;
            check(cghip_host_alloc(dev_->handle(), static_cast<std::size_t>(plane.size), 1, &ptr, &hb), "cghip_host_alloc");
            std::shared_ptr<void> storage(hb, [dev = dev_](void* h) { cghip_host_release(static_cast<cghip_hostbuf*>(h)); });
            image_data.emplace_back(static_cast<std::uint8_t*>(ptr), static_cast<std::size_t>(plane.size), storage);
        }
        std::weak_ptr<impl> weak_self = shared_from_this();
        return core::mutable_frame(
            tag, std::move(image_data), desc,
            [weak_self, desc](std::vector<array<const std::uint8_t>> image_data) -> std::shared_ptr<void> {
                auto self = weak_self.lock();
                if (!self)
                    return nullptr;
                return self->upload(desc, [&](int n) { return image_data[static_cast<std::size_t>(n)]; });
            });
    }
};

image_mixer::image_mixer(const std::shared_ptr<device>& dev, int channel_id)
    : impl_(std::make_shared<impl>(dev, channel_id))
{
}
image_mixer::~image_mixer() {}
void image_mixer::push(const core::frame_transform& transform) { impl_->push(transform); }
void image_mixer::visit(const core::const_frame& frame) { impl_->visit(frame); }
void image_mixer::pop() { impl_->pop(); }
std::future<array<const std::uint8_t>> image_mixer::operator()(const core::video_format_desc& format_desc)
{
    return impl_->render(format_desc);
}
core::mutable_frame image_mixer::create_frame(const void* tag, const core::pixel_format_desc& desc)
{
    return impl_->create_frame(tag, desc);
}
std::vector<draw_call> image_mixer::flatten(const core::video_format_desc& format_desc, std::vector<cghip_surface>* surfaces)
{
    return impl_->flatten(format_desc, surfaces);
}
void image_mixer::render_into(const core::video_format_desc& format_desc, void* device_ptr, std::size_t bytes)
{
    impl_->render_into(format_desc, device_ptr, bytes);
}
float image_mixer::last_kernel_ms() const { return impl_->last_ms_; }
void  image_mixer::set_readback(bool on) { impl_->readback_ = on; }
void  image_mixer::set_field_rows_only(bool on) { impl_->field_rows_only_ = on; }
void  image_mixer::set_kernel_timing(bool on) { impl_->kernel_timing_ = on; }

/* ------------------------------------------------------------------ accelerator */

struct accelerator::impl
{
    std::mutex                             mutex_;
    std::map<int, std::shared_ptr<device>> devices_;
    int                                    count_ = -1;
};

accelerator::accelerator()
    : impl_(new impl())
{
}
accelerator::~accelerator() {}

int accelerator::device_count() const
{
    std::lock_guard<std::mutex> lock(impl_->mutex_);
    if (impl_->count_ < 0) {
        int n = 0;
        if (cghip_device_count(&n) != CGHIP_OK)
            n = 0;
        impl_->count_ = n;
    }
    return impl_->count_;
}

std::shared_ptr<device> accelerator::get_device(int gpu_index) /* accelerator.cpp:27-34 */
{
    std::lock_guard<std::mutex> lock(impl_->mutex_);
    auto&                       d = impl_->devices_[gpu_index];
    if (!d)
        d = std::make_shared<device>(gpu_index);
    return d;
}

std::unique_ptr<image_mixer> accelerator::create_image_mixer(int channel_id) /* accelerator.cpp:22-25 */
{
    const int n = device_count();
    if (n <= 0)
        throw not_supported("no HIP device: the accelerator/hip image mixer needs a gfx950 GPU");
    /* per-channel device placement: channels are independent, so shard them round-robin */
    return create_image_mixer(channel_id, ((channel_id % n) + n) % n);
}

std::unique_ptr<image_mixer> accelerator::create_image_mixer(int channel_id, int gpu_index)
{
    return std::make_unique<image_mixer>(get_device(gpu_index), channel_id);
}

}} // namespace accelerator::hip
} // namespace caspar
