/*
 * accelerator/hip/image_mixer.h -- the MI355X image mixer as a core::image_mixer, for the CasparCG Server 2.3.x tree.
 *
 * ADAPTER SOURCE, NOT COMPILED IN THIS REPOSITORY: it includes the reference's own headers (common/, core/, boost),
 * which this build image does not have. Drop the directory into src/accelerator/hip/, add the two .cpp files and
 * libcghip.so to src/accelerator/CMakeLists.txt, apply accelerator.cpp.diff. The same logic, against restated value
 * types, is compiled and tested here as server_amd/host/image_mixer.{h,cpp} (INTEGRATION.md lists every difference).
 *
 * Mirrors accelerator/ogl/image/image_mixer.h:33-67 member for member; everything GPU-side goes through the C ABI of
 * include/cghip.h.
 */
#pragma once

#include <common/array.h>
#include <common/memory.h>

#include <core/frame/frame.h>
#include <core/mixer/image/image_mixer.h>
#include <core/video_format.h>

#include <future>
#include <memory>

namespace caspar { namespace accelerator { namespace hip {

class image_mixer final : public core::image_mixer
{
  public:
    image_mixer(const spl::shared_ptr<class device>& dev, int channel_id);
    image_mixer(const image_mixer&) = delete;
    ~image_mixer();
    image_mixer& operator=(const image_mixer&) = delete;

    std::future<array<const std::uint8_t>> operator()(const core::video_format_desc& format_desc) override;
    core::mutable_frame                    create_frame(const void* tag, const core::pixel_format_desc& desc) override;

    // core::image_mixer (frame_visitor)
    void push(const core::frame_transform& frame) override;
    void visit(const core::const_frame& frame) override;
    void pop() override;

    /* Extensions the unchanged caller never uses (core/mixer/mixer.cpp:78-79 builds an 8-bit bgra desc):
       16-bit BGRA output and an in-kernel field mask. See INTEGRATION.md section 2a for where a channel would set them. */
    void set_output_bits(int bits);       // 8 (default, the reference's output) or 16
    void set_field_mode(int field_mode);  // 0 progressive (default), 1 / 2: only even / odd rows are drawn

  private:
    struct impl;
    std::shared_ptr<impl> impl_;
};

}}} // namespace caspar::accelerator::hip
