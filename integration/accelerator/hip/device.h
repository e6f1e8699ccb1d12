/*
 * accelerator/hip/device.h -- one MI355X behind the reference's accelerator_device interface
 * (accelerator/accelerator.h:13-18: wptree info() const, future<void> gc()), the counterpart of
 * accelerator/ogl/util/device.h. ADAPTER SOURCE, NOT COMPILED IN THIS REPOSITORY (see image_mixer.h).
 */
#pragma once

#include "../accelerator.h"

#include <boost/property_tree/ptree.hpp>

#include <future>
#include <memory>

struct cghip_device;

namespace caspar { namespace accelerator { namespace hip {

class device final
    : public std::enable_shared_from_this<device>
    , public accelerator_device
{
  public:
    explicit device(int gpu_index); // throws not_supported without a usable gfx950 GPU (ogl: device.cpp:109-114)
    device(const device&) = delete;
    ~device();
    device& operator=(const device&) = delete;

    boost::property_tree::wptree info() const override; // GL INFO (protocol/amcp/AMCPCommandsImpl.cpp:1456-1469)
    std::future<void>            gc() override;         // GL GC   (:1471-1481)

    cghip_device* handle() const { return dev_; }
    int           gpu() const { return gpu_; }

    static int count(); // GPUs visible to the process

  private:
    cghip_device* dev_ = nullptr;
    int           gpu_ = 0;
};

}}} // namespace caspar::accelerator::hip
