/*
 * accelerator/hip/image_mixer.cpp -- core::image_mixer on MI355X for the CasparCG Server 2.3.x tree.
 *
 * ADAPTER SOURCE, NOT COMPILED IN THIS REPOSITORY (needs the reference's headers and boost; see image_mixer.h). The
 * compiled and tested twin is server_amd/host/image_mixer.cpp; the two differ only in the types named in
 * INTEGRATION.md section 2a (boost::any opaque, array<> storage, caspar exceptions, spl::shared_ptr).
 *
 * Structure follows accelerator/ogl/image/image_mixer.cpp (GPLv3, (c) Sveriges Television AB) function for function:
 * the transform stack and layer tree (:54-72, :247-299) are the same code because they ARE the interface contract; the
 * renderer (:74-226) keeps its control flow but RECORDS draws instead of issuing GL calls, and one call of
 * cghip_mixer_render_async composites the recorded list in one fused launch per frame.
 */
#include "image_mixer.h"

#include "device.h"

#include <cghip.h>

#include <common/except.h>
#include <common/future.h>
#include <common/log.h>

#include <core/frame/frame_transform.h>
#include <core/frame/geometry.h>
#include <core/frame/pixel_format.h>

#include <boost/any.hpp>

#include <algorithm>
#include <mutex>
#include <vector>

namespace caspar { namespace accelerator { namespace hip {

void check(int rc, const char* what)
{
    if (rc == CGHIP_OK)
        return;
    const std::string msg = std::string(what) + ": " + cghip_last_error();
    switch (rc) {
        case CGHIP_ERR_INVALID: CASPAR_THROW_EXCEPTION(invalid_argument() << msg_info(msg));
        case CGHIP_ERR_NO_DEVICE:
        case CGHIP_ERR_UNSUPPORTED: CASPAR_THROW_EXCEPTION(not_supported() << msg_info(msg));
        case CGHIP_ERR_NOMEM: CASPAR_THROW_EXCEPTION(bad_alloc() << msg_info(msg));
        default: CASPAR_THROW_EXCEPTION(caspar_exception() << msg_info(msg));
    }
}

namespace {

using plane_ptr = std::shared_ptr<cghip_plane>;
plane_ptr own(cghip_plane* p) { return plane_ptr(p, [](cghip_plane* q) { cghip_plane_release(q); }); }

/* What const_frame::opaque() carries for frames made by this mixer's frame factory: the uploaded planes and the pinned
 * arrays they are being copied from. `home` names the GPU; a mixer on another GPU (route_producer.cpp:40-126) gets a
 * device-to-device copy over xGMI, cached here per destination so that N routes of a frame onto a GPU copy it once. */
struct uploaded
{
    spl::shared_ptr<device>                                           home;
    // routed copy: the origin's device and the pinned buffers its upload may still read (the C ABI retains the source
    // planes). Never a pointer to the origin itself: its `routed` cache holds this copy, that would be a cycle.
    std::shared_ptr<device>                                           origin_home;
    std::vector<array<const std::uint8_t>>                            origin_sources;
    std::vector<array<const std::uint8_t>>                            sources; // before planes: destroyed after them
    std::vector<plane_ptr>                                            planes;
    std::mutex                                                        route_mutex;
    std::vector<std::pair<cghip_device*, std::shared_ptr<uploaded>>>  routed;

    explicit uploaded(const spl::shared_ptr<device>& home) : home(home) {}

    static std::shared_ptr<uploaded> on_device(const std::shared_ptr<uploaded>& self, const spl::shared_ptr<device>& dst)
    {
        if (self->home->handle() == dst->handle())
            return self;
        std::lock_guard<std::mutex> lock(self->route_mutex);
        for (auto& r : self->routed)
            if (r.first == dst->handle())
                return r.second;
        auto copy    = std::make_shared<uploaded>(dst);
        copy->origin_home    = self->home;
        copy->origin_sources = self->sources;
        for (auto& pl : self->planes) {
            cghip_plane* out = nullptr;
            check(cghip_plane_peer_copy_async(dst->handle(), pl.get(), &out), "cghip_plane_peer_copy_async");
            copy->planes.push_back(own(out));
        }
        self->routed.emplace_back(dst->handle(), copy);
        return copy;
    }
};
using uploaded_ptr = std::shared_ptr<uploaded>;

struct item // ogl/image/image_mixer.cpp:54-60
{
    core::pixel_format_desc pix_desc = core::pixel_format::invalid;
    uploaded_ptr            textures;
    core::image_transform   transform;
    core::frame_geometry    geometry = core::frame_geometry::get_default();
};

struct layer // :62-72
{
    std::vector<layer> sublayers;
    std::vector<item>  items;
    core::blend_mode   blend_mode;

    explicit layer(core::blend_mode blend_mode) : blend_mode(blend_mode) {}
};

void to_pod(const core::image_transform& t, cghip_transform& o)
{
    o         = cghip_transform{};
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

/* image_renderer (:74-226) as a recorder: the same control flow, but every kernel_.draw becomes one cghip_draw and every
 * create_texture one more surface slot. NB :207-225: `source_buffer` is a named rvalue reference there and
 * spl::make_shared_ptr copies from it, so a layer's mix texture is NOT reset by its composite. */
class draw_recorder
{
    const core::video_format_desc& format_desc_;
    std::vector<cghip_draw>&       draws_;
    std::vector<cghip_surface>&    surfaces_;
    std::vector<plane_ptr>&        keep_;

    static constexpr int none = -1;

  public:
    draw_recorder(const core::video_format_desc& format_desc,
                  std::vector<cghip_draw>&       draws,
                  std::vector<cghip_surface>&    surfaces,
                  std::vector<plane_ptr>&        keep)
        : format_desc_(format_desc), draws_(draws), surfaces_(surfaces), keep_(keep)
    {
        surfaces_.assign(1, cghip_surface{4, 0}); // target_texture :95
    }

    void draw(int target, const std::vector<layer>& layers) // :104-114
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
            CASPAR_THROW_EXCEPTION(not_supported() << msg_info("scene needs more than CGHIP_MAX_SURFACES temporaries"));
        surfaces_.push_back(cghip_surface{channels, 0});
        return static_cast<int>(surfaces_.size()) - 1;
    }

    void draw(int target, const layer& l, int& layer_key) // :116-154
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

    static void set_geometry(cghip_draw& d, const core::frame_geometry& g)
    {
        d.geometry_is_quad = g.type() == core::frame_geometry::geometry_type::quad;
        d.ncoords          = static_cast<int>(std::min<std::size_t>(g.data().size(), 4));
        for (int i = 0; i < d.ncoords; ++i) {
            auto& c     = g.data()[i];
            d.coords[i] = cghip_coord{c.vertex_x, c.vertex_y, c.texture_x, c.texture_y, c.texture_r, c.texture_q};
        }
    }

    void draw(int target, const item& it, int& layer_key, int& local_key, int& local_mix) // :156-205
    {
        cghip_draw d{};
        d.pixel_format   = static_cast<int>(it.pix_desc.format);
        d.nplanes        = static_cast<int>(it.textures ? it.textures->planes.size() : 0);
        d.source_surface = none;
        for (int n = 0; n < d.nplanes && n < 4; ++n) {
            d.planes[n] = it.textures->planes[n].get();
            keep_.push_back(it.textures->planes[n]);
        }
        to_pod(it.transform, d.transform);
        set_geometry(d, it.geometry);
        d.blend_mode   = static_cast<int>(core::blend_mode::normal);
        d.keyer        = CGHIP_KEYER_LINEAR;
        d.aspect_ratio = static_cast<double>(format_desc_.square_width) / static_cast<double>(format_desc_.square_height);

        if (it.transform.is_key) {
            if (local_key == none)
                local_key = create_surface(1);
            d.target = local_key, d.local_key = none, d.layer_key = none;
        } else if (it.transform.is_mix) {
            if (local_mix == none)
                local_mix = create_surface(4);
            d.target    = local_mix;
            d.local_key = local_key;
            local_key   = none; // std::move(local_key_texture) :190
            d.layer_key = layer_key;
            d.keyer     = CGHIP_KEYER_ADDITIVE;
        } else {
            draw(target, local_mix, core::blend_mode::normal);
            d.target    = target;
            d.local_key = local_key;
            local_key   = none; // :200
            d.layer_key = layer_key;
        }
        draws_.push_back(d);
    }

    void draw(int target, int source_buffer, core::blend_mode blend_mode) // :207-225
    {
        if (source_buffer == none)
            return;
        cghip_draw d{};
        d.pixel_format   = CGHIP_PF_BGRA;
        d.source_surface = source_buffer;
        d.target         = target;
        d.local_key = d.layer_key = none;
        d.blend_mode     = static_cast<int>(blend_mode);
        d.keyer          = CGHIP_KEYER_LINEAR;
        to_pod(core::image_transform(), d.transform);
        set_geometry(d, core::frame_geometry::get_default());
        d.aspect_ratio = 1.0;
        draws_.push_back(d);
    }
};

} // namespace

struct image_mixer::impl
    : public core::frame_factory
    , public std::enable_shared_from_this<impl>
{
    spl::shared_ptr<device>            dev_;
    cghip_mixer*                       mixer_ = nullptr;
    std::vector<core::image_transform> transform_stack_;
    std::vector<layer>                 layers_; // layer/stream/items
    std::vector<layer*>                layer_stack_;
    int                                bits_       = 8;
    int                                field_mode_ = 0;

    impl(const spl::shared_ptr<device>& dev, int channel_id)
        : dev_(dev)
        , transform_stack_(1)
    {
        check(cghip_mixer_create(dev_->handle(), channel_id, &mixer_), "cghip_mixer_create");
        CASPAR_LOG(info) << L"Initialized HIP Accelerated GPU Image Mixer for channel " << channel_id << L" on GPU "
                         << dev_->gpu();
    }

    ~impl() { cghip_mixer_destroy(mixer_); }

    void push(const core::frame_transform& transform) // ogl :247-264, unchanged
    {
        auto previous_layer_depth = transform_stack_.back().layer_depth;
        transform_stack_.push_back(transform_stack_.back() * transform.image_transform);
        auto new_layer_depth = transform_stack_.back().layer_depth;

        if (previous_layer_depth < new_layer_depth) {
            layer new_layer(transform_stack_.back().blend_mode);

            if (layer_stack_.empty()) {
                layers_.push_back(std::move(new_layer));
                layer_stack_.push_back(&layers_.back());
            } else {
                layer_stack_.back()->sublayers.push_back(std::move(new_layer));
                layer_stack_.push_back(&layer_stack_.back()->sublayers.back());
            }
        }
    }

    uploaded_ptr upload(const core::pixel_format_desc& desc, const std::vector<array<const std::uint8_t>>& image_data)
    {
        auto up = std::make_shared<uploaded>(dev_);
        for (int n = 0; n < static_cast<int>(desc.planes.size()); ++n) {
            auto&        pl = desc.planes[n];
            cghip_plane* p  = nullptr;
            // uyvy is two views of the same bytes (modules/ffmpeg/util/av_util.cpp:163-171): upload once, alias
            if (n == 1 && desc.format == core::pixel_format::uyvy && desc.planes.size() == 2 &&
                desc.planes[0].size == pl.size)
                check(cghip_plane_view(up->planes[0].get(), pl.width, pl.height, pl.stride, &p), "cghip_plane_view");
            else
                check(cghip_plane_upload_async(dev_->handle(), image_data[n].data(), pl.width, pl.height, pl.stride, 8, &p),
                      "cghip_plane_upload_async");
            up->planes.push_back(own(p));
            up->sources.push_back(image_data[n]);
        }
        return up;
    }

    void visit(const core::const_frame& frame) // ogl :266-293
    {
        if (frame.pixel_format_desc().format == core::pixel_format::invalid)
            return;
        if (frame.pixel_format_desc().planes.empty())
            return;

        item item;
        item.pix_desc  = frame.pixel_format_desc();
        item.transform = transform_stack_.back();
        item.geometry  = frame.geometry();

        // pointer-form any_cast: a frame from another accelerator or with an empty opaque is uploaded here (the value
        // form the ogl mixer uses at :279 would throw bad_any_cast on a foreign type)
        auto textures_ptr = boost::any_cast<uploaded_ptr>(&frame.opaque());
        if (textures_ptr && *textures_ptr && (*textures_ptr)->planes.size() == item.pix_desc.planes.size()) {
            item.textures = uploaded::on_device(*textures_ptr, dev_); // routed from another GPU: one peer copy per GPU
        } else {
            std::vector<array<const std::uint8_t>> data;
            for (std::size_t n = 0; n < item.pix_desc.planes.size(); ++n)
                data.push_back(frame.image_data(n));
            item.textures = upload(item.pix_desc, data);
        }
        layer_stack_.back()->items.push_back(std::move(item));
    }

    void pop() // ogl :295-299
    {
        transform_stack_.pop_back();
        layer_stack_.resize(transform_stack_.back().layer_depth);
    }

    std::future<array<const std::uint8_t>> render(const core::video_format_desc& format_desc) // ogl :301-304, :86-101
    {
        auto layers = std::move(layers_);
        layers_.clear();
        layer_stack_.clear();

        if (layers.empty()) { // Bypass GPU with empty frame. (:89-92)
            static const std::vector<uint8_t> buffer(8192 * 8192 * 8, 0);
            return make_ready_future(array<const std::uint8_t>(buffer.data(), format_desc.size * (bits_ / 8), true));
        }

        std::vector<cghip_draw>    draws;
        std::vector<cghip_surface> surfaces;
        auto                       keep = std::make_shared<std::vector<plane_ptr>>();
        draw_recorder(format_desc, draws, surfaces, *keep).draw(0, layers);

        cghip_format  fmt{format_desc.width, format_desc.height, format_desc.square_width, format_desc.square_height,
                         bits_,             field_mode_};
        cghip_ticket* ticket = nullptr;
        check(cghip_mixer_render_async(mixer_, &fmt, surfaces.data(), static_cast<int>(surfaces.size()), draws.data(),
                                       static_cast<int>(draws.size()), CGHIP_RENDER_READBACK, &ticket),
              "cghip_mixer_render_async");
        // the buffer a consumer holds keeps the ticket (pinned memory back to the pool on release) and the device
        std::shared_ptr<cghip_ticket> owner(ticket, [dev = dev_](cghip_ticket* t) { cghip_ticket_release(t); });

        // deferred like common/future.h:9-13: core::mixer calls get() one tick later (mixer.cpp:85-91); the wait is on
        // a hipEvent, not a 2 ms poll (device.cpp:276-288)
        return std::async(std::launch::deferred, [owner, keep]() mutable -> array<const std::uint8_t> {
            const std::uint8_t* data = nullptr;
            std::size_t         size = 0;
            check(cghip_ticket_wait(owner.get(), &data, &size), "cghip_ticket_wait");
            return array<const std::uint8_t>(data, size, owner); // array.h:120-126: the storage owns the ticket
        });
    }

    core::mutable_frame create_frame(const void* tag, const core::pixel_format_desc& desc) override // ogl :306-331
    {
        std::vector<array<std::uint8_t>> image_data;
        for (auto& plane : desc.planes) {
            void*          ptr = nullptr;
            cghip_hostbuf* hb  = nullptr;
            check(cghip_host_alloc(dev_->handle(), plane.size, 1, &ptr, &hb), "cghip_host_alloc"); // pinned, pooled
            std::shared_ptr<void> storage(hb, [dev = dev_](void* h) { cghip_host_release(static_cast<cghip_hostbuf*>(h)); });
            image_data.emplace_back(static_cast<std::uint8_t*>(ptr), static_cast<std::size_t>(plane.size), storage);
        }

        std::weak_ptr<image_mixer::impl> weak_self = shared_from_this();
        return core::mutable_frame(
            tag,
            std::move(image_data),
            array<int32_t>{},
            desc,
            [weak_self, desc](std::vector<array<const std::uint8_t>> image_data) -> boost::any {
                auto self = weak_self.lock();
                if (!self) {
                    return boost::any{};
                }
                return self->upload(desc, image_data); // H2D starts at commit, long before visit (SURVEY call stack B)
            });
    }
};

image_mixer::image_mixer(const spl::shared_ptr<device>& dev, int channel_id)
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
void image_mixer::set_output_bits(int bits) { impl_->bits_ = bits == 16 ? 16 : 8; }
void image_mixer::set_field_mode(int field_mode) { impl_->field_mode_ = field_mode; }

}}} // namespace caspar::accelerator::hip
