/*
 * accelerator/hip image mixer -- host side above the C ABI (include/cghip.h).
 *
 * Mirrors the reference's plugin interface for this path, same names and argument meaning:
 *   core/mixer/image/image_mixer.h:33-57      core::image_mixer : frame_visitor, frame_factory
 *   core/frame/frame_visitor.h:26-38          push / visit / pop
 *   core/frame/frame_factory.h:32-48          create_frame
 *   accelerator/ogl/image/image_mixer.h       ogl::image_mixer (the class this one replaces)
 *   accelerator/accelerator.h:13-36           accelerator / accelerator_device
 * boost::any, spl::shared_ptr and caspar::array are replaced by std equivalents (no boost in this image);
 * the adapter for the real tree is in INTEGRATION.md.
 */
#pragma once

#include <cstdint>
#include <functional>
#include <future>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cghip.h"
#include "core_types.h"

namespace caspar {

/* common/array.h:12-170 -- pointer + size + type-erased owner */
template <typename T>
class array final
{
  public:
    array() = default;
    array(T* ptr, std::size_t size, std::shared_ptr<void> storage)
        : ptr_(ptr)
        , size_(size)
        , storage_(std::move(storage))
    {
    }
    T*          begin() const { return ptr_; }
    T*          data() const { return ptr_; }
    T*          end() const { return ptr_ + size_; }
    std::size_t size() const { return size_; }
    explicit    operator bool() const { return ptr_ != nullptr; }

    template <typename U>
    array<U> cast() const
    {
        return array<U>(ptr_, size_, storage_);
    }

  private:
    T*                    ptr_  = nullptr;
    std::size_t           size_ = 0;
    std::shared_ptr<void> storage_;
};

struct caspar_exception : std::runtime_error /* common/except.h */
{
    using std::runtime_error::runtime_error;
};
struct invalid_argument : caspar_exception
{
    using caspar_exception::caspar_exception;
};
struct not_supported : caspar_exception
{
    using caspar_exception::caspar_exception;
};

namespace core {

/* core/frame/frame.h: frames as the mixer sees them. opaque() carries the uploaded planes. */
class const_frame;

class mutable_frame final
{
  public:
    using commit_t = std::function<std::shared_ptr<void>(std::vector<array<const std::uint8_t>>)>;

    mutable_frame(const void* tag, std::vector<array<std::uint8_t>> image_data, const pixel_format_desc& desc,
                  commit_t commit)
        : tag_(tag)
        , image_data_(std::move(image_data))
        , desc_(desc)
        , commit_(std::move(commit))
    {
    }
    const core::pixel_format_desc& pixel_format_desc() const { return desc_; }
    array<std::uint8_t>&           image_data(std::size_t index) { return image_data_.at(index); }
    core::frame_geometry&          geometry() { return geometry_; }
    const void*                    stream_tag() const { return tag_; }

  private:
    friend class const_frame;
    const void*                      tag_;
    std::vector<array<std::uint8_t>> image_data_;
    core::pixel_format_desc          desc_;
    core::frame_geometry             geometry_ = frame_geometry::get_default();
    commit_t                         commit_;
};

class const_frame final
{
  public:
    const_frame() = default;
    /* frame.cpp:122-135: converting a mutable_frame runs the commit callback (starts the upload) */
    explicit const_frame(mutable_frame&& other)
        : desc_(other.desc_)
        , geometry_(other.geometry_)
        , tag_(other.tag_)
    {
        for (auto& a : other.image_data_)
            image_data_.push_back(a.cast<const std::uint8_t>());
        if (other.commit_)
            opaque_ = other.commit_(image_data_);
    }
    /* frames that did not come from this mixer's create_frame (foreign or empty opaque) */
    const_frame(std::vector<array<const std::uint8_t>> image_data, const core::pixel_format_desc& desc)
        : image_data_(std::move(image_data))
        , desc_(desc)
    {
    }
    const core::pixel_format_desc&    pixel_format_desc() const { return desc_; }
    const array<const std::uint8_t>&  image_data(std::size_t index) const { return image_data_.at(index); }
    const core::frame_geometry&       geometry() const { return geometry_; }
    core::frame_geometry&             geometry() { return geometry_; }
    const std::shared_ptr<void>&      opaque() const { return opaque_; }
    const void*                       stream_tag() const { return tag_; }

  private:
    std::vector<array<const std::uint8_t>> image_data_;
    core::pixel_format_desc                desc_ = core::pixel_format::invalid;
    core::frame_geometry                   geometry_ = frame_geometry::get_default();
    const void*                            tag_      = nullptr;
    std::shared_ptr<void>                  opaque_;
};

} // namespace core

namespace accelerator { namespace hip {

/* accelerator_device (accelerator/accelerator.h:13-18): info() / gc() behind GL INFO / GL GC */
class device final
{
  public:
    explicit device(int gpu_index);
    ~device();
    device(const device&) = delete;
    device& operator=(const device&) = delete;

    std::string       info() const;
    std::future<void> gc();
    cghip_device*     handle() const { return dev_; }
    int               gpu_index() const { return gpu_; }

  private:
    cghip_device* dev_ = nullptr;
    int           gpu_ = 0;
};

/* One flattened kernel_.draw() call of the reference renderer, before it crosses the C ABI. Exposed so
 * host-logic tests can compare the draw list with the oracle's. */
struct draw_call
{
    cghip_draw                                 draw;
    std::vector<std::shared_ptr<cghip_plane>>  keep; /* planes referenced by draw.planes[] */
};

class image_mixer final
{
  public:
    image_mixer(const std::shared_ptr<device>& dev, int channel_id);
    ~image_mixer();
    image_mixer(const image_mixer&) = delete;
    image_mixer& operator=(const image_mixer&) = delete;

    /* core::frame_visitor */
    void push(const core::frame_transform& transform);
    void visit(const core::const_frame& frame);
    void pop();

    /* core::image_mixer::operator() -- returns the BGRA frame, row 0 = top, premultiplied */
    std::future<array<const std::uint8_t>> operator()(const core::video_format_desc& format_desc);

    /* core::frame_factory */
    core::mutable_frame create_frame(const void* tag, const core::pixel_format_desc& desc);

    /* Introspection (not in the reference interface): flatten the pending scene into the draw list the
     * renderer would issue, without rendering; and device-side timing of the last rendered frame. */
    std::vector<draw_call> flatten(const core::video_format_desc& format_desc, std::vector<cghip_surface>* surfaces);
    /* Not in the reference interface: composites the pending scene straight into device memory of the caller
       (format_desc.size bytes on this mixer's GPU) through the entry point that allocates nothing
       (cghip_mixer_render_noalloc, include/cghip.h); returns when the frame is complete. */
    void                   render_into(const core::video_format_desc& format_desc, void* device_ptr, std::size_t bytes);
    float                  last_kernel_ms() const;
    void                   set_readback(bool on); /* false: frames stay in HBM (kernel-only benchmarking) */
    /* interlaced channels (format_desc.field_mode 1 / 2): read back only the rows of the rendered field; the
       consumer weaves two such frames (decklink_consumer.cpp:293-308). Halves the D2H traffic. */
    void                   set_field_rows_only(bool on);
    void                   set_kernel_timing(bool on); /* false: no hipEvent pair per frame, last_kernel_ms() keeps its last measured value */

  private:
    struct impl;
    std::shared_ptr<impl> impl_;
};

/* accelerator (accelerator/accelerator.h:20-36, accelerator.cpp:22-34): lazily creates one device per GPU
 * and places channel c on GPU (c % device_count) unless told otherwise. */
class accelerator final
{
  public:
    accelerator();
    ~accelerator();
    std::unique_ptr<image_mixer> create_image_mixer(int channel_id);
    std::unique_ptr<image_mixer> create_image_mixer(int channel_id, int gpu_index);
    std::shared_ptr<device>      get_device(int gpu_index = 0);
    int                          device_count() const;

  private:
    struct impl;
    std::unique_ptr<impl> impl_;
};

}} // namespace accelerator::hip
} // namespace caspar
