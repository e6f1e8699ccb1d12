/* accelerator/hip/device.cpp -- ADAPTER SOURCE, NOT COMPILED IN THIS REPOSITORY (see image_mixer.h). */
#include "device.h"

#include <cghip.h>

#include <common/except.h>
#include <common/log.h>

#include <boost/property_tree/json_parser.hpp>

#include <sstream>

namespace caspar { namespace accelerator { namespace hip {

void check(int rc, const char* what); // image_mixer.cpp

device::device(int gpu_index)
    : gpu_(gpu_index)
{
    const int rc = cghip_device_create(gpu_index, &dev_);
    if (rc == CGHIP_ERR_NO_DEVICE)
        CASPAR_THROW_EXCEPTION(not_supported() << msg_info("accelerator/hip needs a gfx950 (MI355X) GPU: " +
                                                           std::string(cghip_last_error())));
    check(rc, "cghip_device_create");
    CASPAR_LOG(info) << L"Initialized HIP device for GPU " << gpu_index;
}

device::~device() { cghip_device_destroy(dev_); }

int device::count()
{
    int n = 0;
    return cghip_device_count(&n) == CGHIP_OK ? n : 0;
}

boost::property_tree::wptree device::info() const
{
    // cghip_device_info writes one flat JSON object (pool sizes, live planes / tickets, frames rendered, kernel
    // launches, peer copies and which peers granted direct access); GL INFO prints the tree as XML
    char json[4096];
    check(cghip_device_info(dev_, json, sizeof json), "cghip_device_info");
    std::wistringstream            in(u16(std::string(json)));
    boost::property_tree::wptree   tree, info;
    boost::property_tree::read_json(in, tree);
    info.add_child(L"hip.device", tree);
    return info;
}

std::future<void> device::gc()
{
    // the ogl device posts its gc to the GL thread (device.cpp:418-436); here the pools are mutex-protected and the
    // call synchronises the GPU itself, so a deferred future run by the AMCP thread is enough
    auto self = shared_from_this();
    return std::async(std::launch::deferred, [self] { check(cghip_device_gc(self->dev_), "cghip_device_gc"); });
}

}}} // namespace caspar::accelerator::hip
