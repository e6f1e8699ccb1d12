/*
 * cghip_runtime.cpp -- device, pools, planes, mixers and tickets behind the C ABI (include/cghip.h).
 *
 * HIP counterpart of accelerator/ogl/util/device.cpp: where the reference funnels every channel through
 * ONE GL context on ONE thread (device.cpp:82-142) and polls a fence every 2 ms for readback
 * (:264-302), this runtime gives each device an upload stream shared by producer threads, each mixer
 * (= channel) its own compute/readback stream, and completes frames with hipEvents. Pools are keyed by
 * byte size like the reference's (device.cpp:194-233) and grow instead of blocking.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "cghip_internal.h"

namespace cghip {

static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }
int  fail(int status, const std::string& msg)
{
    g_err = msg;
    return status;
}
int fail_hip(hipError_t e, const char* what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return CGHIP_ERR_HIP;
}

static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* layout of the scratch array of cghip_mixer_render_noalloc: [ MAX_OPS KOps | op masks | hand-over surfaces ] */
constexpr size_t NOALLOC_MASK_REGION = 512u << 10;
static size_t    noalloc_ops_bytes() { return round_up((size_t)CGHIP_NOALLOC_MAX_OPS * sizeof(KOp), 256); }

} // namespace cghip

using namespace cghip;

/* ------------------------------------------------------------------ pools */

int cghip_device::event_get(int kind, hipEvent_t* out)
{
    {
        std::lock_guard<std::mutex> g(mu_ev);
        if (!event_pool[kind].empty()) {
            *out = event_pool[kind].back();
            event_pool[kind].pop_back();
            return CGHIP_OK;
        }
    }
    CGHIP_TRY_HIP(hipSetDevice(gpu));
    /* CGHIP_EVENT_WAIT=spin: busy-waiting completion events (lowest wake-up latency, one core per waiting thread) */
    static const bool spin = [] { const char* v = getenv("CGHIP_EVENT_WAIT"); return v && std::strcmp(v, "spin") == 0; }();
    const unsigned    block = spin ? 0u : (unsigned)hipEventBlockingSync;
    const unsigned    flags = kind == EV_TIMED  ? block
                              : kind == EV_DONE ? (block | hipEventDisableTiming)
                                                : hipEventDisableTiming;
    CGHIP_TRY_HIP(hipEventCreateWithFlags(out, flags));
    events_created++;
    return CGHIP_OK;
}
void cghip_device::event_put(int kind, hipEvent_t ev)
{
    if (!ev)
        return;
    std::lock_guard<std::mutex> g(mu_ev);
    event_pool[kind].push_back(ev);
}

int cghip_device::dev_alloc(size_t bytes, void** out)
{
    bytes = round_up(bytes ? bytes : 1, 256);
    {
        std::lock_guard<std::mutex> g(mu_dev);
        auto                        it = device_pool.free_.find(bytes);
        if (it != device_pool.free_.end()) {
            *out = it->second;
            device_pool.free_.erase(it);
            device_pool.bytes_free -= bytes;
            return CGHIP_OK;
        }
    }
    CGHIP_TRY_HIP(hipSetDevice(gpu));
    CGHIP_TRY_HIP(hipMalloc(out, bytes));
    std::lock_guard<std::mutex> g(mu_dev);
    device_pool.bytes_total += bytes;
    device_pool.count_total++;
    return CGHIP_OK;
}
void cghip_device::dev_free(void* p, size_t bytes)
{
    if (!p)
        return;
    bytes = round_up(bytes ? bytes : 1, 256);
    std::lock_guard<std::mutex> g(mu_dev);
    device_pool.free_.emplace(bytes, p);
    device_pool.bytes_free += bytes;
}
int cghip_device::pinned_alloc(size_t bytes, void** out)
{
    bytes = round_up(bytes ? bytes : 1, 4096);
    {
        std::lock_guard<std::mutex> g(mu_pin);
        auto                        it = pinned_pool.free_.find(bytes);
        if (it != pinned_pool.free_.end()) {
            *out = it->second;
            pinned_pool.free_.erase(it);
            pinned_pool.bytes_free -= bytes;
            return CGHIP_OK;
        }
    }
    CGHIP_TRY_HIP(hipSetDevice(gpu));
    CGHIP_TRY_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    std::lock_guard<std::mutex> g(mu_pin);
    pinned_pool.bytes_total += bytes;
    pinned_pool.count_total++;
    return CGHIP_OK;
}
void cghip_device::pinned_free(void* p, size_t bytes)
{
    if (!p)
        return;
    bytes = round_up(bytes ? bytes : 1, 4096);
    std::lock_guard<std::mutex> g(mu_pin);
    pinned_pool.free_.emplace(bytes, p);
    pinned_pool.bytes_free += bytes;
}

/* ------------------------------------------------------------------ C ABI */

extern "C" {

const char* cghip_last_error(void) { return g_err.c_str(); }
int         cghip_abi_version(void) { return CGHIP_ABI_VERSION; }

int cghip_device_count(int* count)
{
    if (!count)
        return fail(CGHIP_ERR_INVALID, "count is null");
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        (void)hipGetLastError();
        return fail_hip(e, "hipGetDeviceCount");
    }
    *count = n;
    return CGHIP_OK;
}

int cghip_device_create(int gpu_index, cghip_device** out)
{
    if (!out)
        return fail(CGHIP_ERR_INVALID, "out is null");
    *out  = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return fail(CGHIP_ERR_NO_DEVICE, "no HIP device visible (the GL reference throws not_supported here)");
    }
    if (gpu_index < 0 || gpu_index >= n)
        return fail(CGHIP_ERR_INVALID, "gpu_index out of range");
    CGHIP_TRY_HIP(hipSetDevice(gpu_index));
    hipDeviceProp_t prop;
    CGHIP_TRY_HIP(hipGetDeviceProperties(&prop, gpu_index));
    cghip_device* d = new cghip_device();
    d->gpu          = gpu_index;
    d->name         = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    hipError_t e    = hipStreamCreateWithFlags(&d->upload_stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        e = hipStreamCreateWithFlags(&d->peer_stream, hipStreamNonBlocking);
        if (e != hipSuccess)
            (void)hipStreamDestroy(d->upload_stream);
    }
    if (e != hipSuccess) {
        delete d;
        return fail_hip(e, "hipStreamCreateWithFlags");
    }
    /* set-up of the allocation-free entry point: this GPU's scratch array address, looked up once (no allocation) */
    void* scratch = nullptr;
    e             = cghip_scratch_address(gpu_index, &scratch, nullptr);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(d->upload_stream);
        (void)hipStreamDestroy(d->peer_stream);
        delete d;
        return fail_hip(e, "hipGetSymbolAddress(scratch)");
    }
    d->noalloc_staging_bytes = noalloc_ops_bytes() + NOALLOC_MASK_REGION;
    e = hipHostMalloc(&d->noalloc_staging, d->noalloc_staging_bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(d->upload_stream);
        (void)hipStreamDestroy(d->peer_stream);
        delete d;
        return fail_hip(e, "hipHostMalloc(noalloc staging)");
    }
    *out = d;
    return CGHIP_OK;
}

int cghip_device_gc(cghip_device* dev)
{ /* device::gc (device.cpp:418-436): drop everything the pools hold */
    if (!dev)
        return fail(CGHIP_ERR_INVALID, "dev is null");
    std::multimap<size_t, void*> d, p;
    std::vector<hipEvent_t>      evs;
    {
        std::lock_guard<std::mutex> g(dev->mu_dev);
        d.swap(dev->device_pool.free_);
        dev->device_pool.bytes_total -= dev->device_pool.bytes_free;
        dev->device_pool.count_total -= d.size();
        dev->device_pool.bytes_free = 0;
    }
    {
        std::lock_guard<std::mutex> g(dev->mu_pin);
        p.swap(dev->pinned_pool.free_);
        dev->pinned_pool.bytes_total -= dev->pinned_pool.bytes_free;
        dev->pinned_pool.count_total -= p.size();
        dev->pinned_pool.bytes_free = 0;
    }
    {
        std::lock_guard<std::mutex> g(dev->mu_ev);
        for (auto& pool : dev->event_pool) {
            evs.insert(evs.end(), pool.begin(), pool.end());
            pool.clear();
        }
    }
    CGHIP_TRY_HIP(hipSetDevice(dev->gpu));
    CGHIP_TRY_HIP(hipDeviceSynchronize());
    for (auto& kv : d)
        (void)hipFree(kv.second);
    for (auto& kv : p)
        (void)hipHostFree(kv.second);
    for (hipEvent_t e : evs)
        (void)hipEventDestroy(e);
    return CGHIP_OK;
}

int cghip_device_destroy(cghip_device* dev)
{
    if (!dev)
        return CGHIP_OK;
    int rc = cghip_device_gc(dev);
    if (dev->upload_stream)
        (void)hipStreamDestroy(dev->upload_stream);
    if (dev->peer_stream)
        (void)hipStreamDestroy(dev->peer_stream);
    if (dev->noalloc_staging)
        (void)hipHostFree(dev->noalloc_staging);
    delete dev;
    return rc;
}

int cghip_device_info(cghip_device* dev, char* json, size_t capacity)
{ /* device::info (device.cpp:335-416) as JSON instead of a wptree */
    if (!dev || !json || capacity == 0)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    size_t free_b = 0, total_b = 0;
    (void)hipSetDevice(dev->gpu);
    (void)hipMemGetInfo(&free_b, &total_b);
    std::lock_guard<std::mutex> g(dev->mu);
    std::lock_guard<std::mutex> g1(dev->mu_dev);
    std::lock_guard<std::mutex> g2(dev->mu_pin);
    std::string peers = "{";
    for (auto& kv : dev->peer_access) {
        if (peers.size() > 1)
            peers += ",";
        peers += "\"" + std::to_string(kv.first) + "\":\"" +
                 (kv.second == 1 ? "granted" : kv.second == 2 ? "same_gpu" : kv.second == -1 ? "refused" : "unknown") + "\"";
    }
    peers += "}";
    int n = snprintf(json, capacity,
                     "{\"gpu\":%d,\"name\":\"%s\",\"hbm_total_bytes\":%zu,\"hbm_free_bytes\":%zu,"
                     "\"device_pool\":{\"blocks\":%zu,\"bytes\":%zu,\"bytes_pooled\":%zu},"
                     "\"pinned_pool\":{\"blocks\":%zu,\"bytes\":%zu,\"bytes_pooled\":%zu},"
                     "\"planes_live\":%lld,\"tickets_live\":%lld,\"mixers\":%lld,\"frames_rendered\":%lld,\"kernel_launches\":%lld,"
                     "\"peer_copies\":%lld,\"peer_copy_bytes\":%lld,\"uploads\":%lld,\"upload_bytes\":%lld,\"events_created\":%lld,\"peer_access\":%s}",
                     dev->gpu, dev->name.c_str(), total_b, free_b, dev->device_pool.count_total,
                     dev->device_pool.bytes_total, dev->device_pool.bytes_free, dev->pinned_pool.count_total,
                     dev->pinned_pool.bytes_total, dev->pinned_pool.bytes_free, (long long)dev->planes_live.load(),
                     (long long)dev->tickets_live.load(), (long long)dev->mixers.load(),
                     (long long)dev->frames_rendered.load(), (long long)dev->kernel_launches.load(),
                     (long long)dev->peer_copies.load(),
                     (long long)dev->peer_copy_bytes.load(), (long long)dev->uploads.load(),
                     (long long)dev->upload_bytes.load(), (long long)dev->events_created.load(), peers.c_str());
    if (n < 0 || (size_t)n >= capacity)
        return fail(CGHIP_ERR_INVALID, "info buffer too small");
    return CGHIP_OK;
}

int cghip_device_synchronize(cghip_device* dev)
{
    if (!dev)
        return fail(CGHIP_ERR_INVALID, "dev is null");
    CGHIP_TRY_HIP(hipSetDevice(dev->gpu));
    CGHIP_TRY_HIP(hipDeviceSynchronize());
    return CGHIP_OK;
}

/* ---- pinned host staging */

int cghip_host_alloc(cghip_device* dev, size_t bytes, int write, void** ptr, cghip_hostbuf** out)
{
    (void)write; /* the reference keeps separate read/write pools because PBOs are mapped one way; pinned memory is both */
    if (!dev || !ptr || !out)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    void* p  = nullptr;
    int   rc = dev->pinned_alloc(bytes, &p);
    if (rc != CGHIP_OK)
        return rc;
    *out = new cghip_hostbuf{dev, p, bytes};
    *ptr = p;
    return CGHIP_OK;
}

int cghip_host_release(cghip_hostbuf* buf)
{
    if (!buf)
        return CGHIP_OK;
    buf->dev->pinned_free(buf->ptr, buf->bytes);
    delete buf;
    return CGHIP_OK;
}

/* ---- planes */

static int plane_new(cghip_device* dev, int width, int height, int channels, int bits, cghip_plane** out)
{
    if (!dev || !out)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    if (width <= 0 || height <= 0 || channels < 1 || channels > 4 || (bits != 8 && bits != 16))
        return fail(CGHIP_ERR_INVALID, "plane needs width,height > 0, 1..4 channels, 8 or 16 bits (device.cpp:196-197)");
    /* the kernels address texels with 32-bit byte offsets from the plane (or row) base */
    if ((unsigned long long)width * (unsigned long long)height * channels * (bits / 8) > 0xFFFFFFF0ull)
        return fail(CGHIP_ERR_UNSUPPORTED, "plane larger than 4 GiB");
    cghip_plane* p = new cghip_plane();
    p->dev = dev, p->width = width, p->height = height, p->channels = channels, p->bits = bits;
    p->bytes = (size_t)width * height * channels * (bits / 8);
    int rc   = dev->dev_alloc(p->bytes + 16, &p->dptr); /* +16: vector loads may touch past the last texel */
    if (rc != CGHIP_OK) {
        delete p;
        return rc;
    }
    rc = dev->event_get(cghip_device::EV_ORDER, &p->ready);
    if (rc != CGHIP_OK) {
        dev->dev_free(p->dptr, p->bytes + 16);
        delete p;
        return rc;
    }
    dev->planes_live++;
    *out = p;
    return CGHIP_OK;
}

int cghip_plane_upload_async(cghip_device* dev, const void* src, int width, int height, int channels, int bits,
                             cghip_plane** out)
{
    if (!src)
        return fail(CGHIP_ERR_INVALID, "src is null");
    cghip_plane* p  = nullptr;
    int          rc = plane_new(dev, width, height, channels, bits, &p);
    if (rc != CGHIP_OK)
        return rc;
    hipError_t e = hipSetDevice(dev->gpu);
    if (e == hipSuccess)
        e = hipMemcpyAsync(p->dptr, src, p->bytes, hipMemcpyHostToDevice, dev->upload_stream);
    if (e == hipSuccess)
        e = hipEventRecord(p->ready, dev->upload_stream);
    if (e != hipSuccess) {
        cghip_plane_release(p);
        return fail_hip(e, "plane upload");
    }
    dev->uploads++;
    dev->upload_bytes += (int64_t)p->bytes;
    *out = p;
    return CGHIP_OK;
}

int cghip_plane_retain(cghip_plane* plane)
{
    if (!plane)
        return fail(CGHIP_ERR_INVALID, "plane is null");
    plane->refs++;
    return CGHIP_OK;
}

int cghip_plane_release(cghip_plane* plane)
{
    if (!plane)
        return CGHIP_OK;
    if (--plane->refs > 0)
        return CGHIP_OK;
    if (plane->parent) { /* a view owns nothing but its reference to the parent */
        cghip_plane* parent = plane->parent;
        plane->dev->planes_live--;
        delete plane;
        return cghip_plane_release(parent);
    }
    cghip_plane* origin = plane->origin;
    if (!plane->stub) {
        /* the block goes back to the pool; whoever takes it next writes on a stream that is ordered after
           every reader through the ready event chain, or after the ticket that retained the plane */
        if (plane->ready) {
            (void)hipEventSynchronize(plane->ready);
            plane->dev->event_put(cghip_device::EV_ORDER, plane->ready);
        }
        plane->dev->dev_free(plane->dptr, plane->bytes + 16);
        plane->dev->planes_live--;
    }
    delete plane;
    return origin ? cghip_plane_release(origin) : CGHIP_OK;
}

int cghip_plane_desc(const cghip_plane* plane, int* width, int* height, int* channels, int* bits, int* gpu_index)
{
    if (!plane)
        return fail(CGHIP_ERR_INVALID, "plane is null");
    if (width) *width = plane->width;
    if (height) *height = plane->height;
    if (channels) *channels = plane->channels;
    if (bits) *bits = plane->bits;
    if (gpu_index) *gpu_index = plane->dev ? plane->dev->gpu : -1;
    return CGHIP_OK;
}

int cghip_plane_peer_copy_async(cghip_device* dst_dev, cghip_plane* src, cghip_plane** out)
{
    if (!dst_dev || !src || !out || src->stub)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    if (src->dev == dst_dev) { /* same GPU: share, like the reference's single GL context */
        cghip_plane_retain(src);
        *out = src;
        return CGHIP_OK;
    }
    cghip_plane* p  = nullptr;
    int          rc = plane_new(dst_dev, src->width, src->height, src->channels, src->bits, &p);
    if (rc != CGHIP_OK)
        return rc;
    hipError_t e = hipSetDevice(dst_dev->gpu);
    if (e == hipSuccess) { /* first plane from that GPU: ask for direct access to its memory */
        std::lock_guard<std::mutex> g(dst_dev->mu);
        int& st = dst_dev->peer_access[src->dev->gpu];
        if (st == 0) {
            if (src->dev->gpu == dst_dev->gpu) {
                st = 2;
            } else {
                int        can = 0;
                hipError_t pe  = hipDeviceCanAccessPeer(&can, dst_dev->gpu, src->dev->gpu);
                if (pe == hipSuccess && can) {
                    pe = hipDeviceEnablePeerAccess(src->dev->gpu, 0);
                    st = (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) ? 1 : -1;
                } else {
                    st = -1;
                }
                (void)hipGetLastError(); /* a refusal is not an error: the copy below is then staged by the runtime */
            }
        }
    }
    if (e == hipSuccess)
        e = hipStreamWaitEvent(dst_dev->peer_stream, src->ready, 0);
    if (e == hipSuccess)
        e = hipMemcpyPeerAsync(p->dptr, dst_dev->gpu, src->dptr, src->dev->gpu, src->bytes, dst_dev->peer_stream);
    if (e == hipSuccess)
        e = hipEventRecord(p->ready, dst_dev->peer_stream);
    if (e != hipSuccess) {
        cghip_plane_release(p);
        return fail_hip(e, "peer copy");
    }
    /* the copy is only enqueued: the source bytes must outlive it, so the copy holds its source plane */
    cghip_plane_retain(src);
    p->origin = src;
    dst_dev->peer_copies++;
    dst_dev->peer_copy_bytes += (int64_t)src->bytes;
    *out = p;
    return CGHIP_OK;
}

int cghip_plane_view(cghip_plane* src, int width, int height, int channels, cghip_plane** out)
{
    if (!src || !out || src->stub)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    if (width <= 0 || height <= 0 || channels < 1 || channels > 4 ||
        (size_t)width * height * channels * (src->bits / 8) != src->bytes)
        return fail(CGHIP_ERR_INVALID, "a view must cover exactly the parent's bytes");
    cghip_plane* root = src->parent ? src->parent : src;
    cghip_plane* p    = new cghip_plane();
    p->dev = src->dev, p->dptr = root->dptr, p->bytes = root->bytes, p->ready = root->ready;
    p->width = width, p->height = height, p->channels = channels, p->bits = src->bits;
    p->parent = root;
    cghip_plane_retain(root);
    p->dev->planes_live++;
    *out = p;
    return CGHIP_OK;
}

/* planner tests: a plane descriptor with no device memory (never rendered) */
int cghip_debug_plane_stub(int width, int height, int channels, int bits, cghip_plane** out)
{
    if (!out)
        return fail(CGHIP_ERR_INVALID, "out is null");
    cghip_plane* p = new cghip_plane();
    p->width = width, p->height = height, p->channels = channels, p->bits = bits;
    p->bytes = (size_t)width * height * channels * (bits / 8);
    p->stub  = true;
    *out     = p;
    return CGHIP_OK;
}

/* ---- mixer */

int cghip_mixer_create(cghip_device* dev, int channel_id, cghip_mixer** out)
{
    if (!dev || !out)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    CGHIP_TRY_HIP(hipSetDevice(dev->gpu));
    cghip_mixer* m = new cghip_mixer();
    m->dev = dev, m->channel_id = channel_id;
    /* Stream priorities (opt-in: CGHIP_STREAM_PRIORITIES=1). The frame's D2H copy is a shader copy on this platform
       (`__amd_rocclr_copyBuffer` in the kernel trace, not an SDMA transfer) bound by PCIe: 0.59 ms for a 33 MB frame. It
       needs few waves but it needs them while the next frame composites, and the compositor fills every wave slot of the
       chip (7 x 72 VGPRs per SIMD): with two plain streams the copy waits for slots, and on some boxes a channel with
       readback runs at nearly the serial sum of kernel and copy (config 3 8-bit 0.93 ms per frame, 16-bit 1.44; other
       boxes 0.69 / 1.25). With the compositor's stream at the lowest priority and the readback stream at the highest the
       same box gave 0.69 and 1.24 ms = 86 % and 94 % of the copy alone (tests/gpu_readback_prio.sh). NOT the default:
       with priorities on, one render of the random-scene suite (tests/test_fuzz.py) never completed - a frame's event
       was never signalled - and the cause inside the runtime's queue scheduling was not found in round 3. */
    int         least = 0, greatest = 0;
    const char* sp    = std::getenv("CGHIP_STREAM_PRIORITIES");
    const bool  prio  = sp != nullptr && std::atoi(sp) != 0 &&
                       hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least;
    hipError_t  e     = prio ? hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, least)
                             : hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        e = hipStreamCreateWithFlags(&m->ops_stream, hipStreamNonBlocking);
        if (e != hipSuccess)
            (void)hipStreamDestroy(m->stream);
    }
    if (e == hipSuccess) {
        e = prio ? hipStreamCreateWithPriority(&m->readback_stream, hipStreamNonBlocking, greatest)
                 : hipStreamCreateWithFlags(&m->readback_stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            (void)hipStreamDestroy(m->ops_stream);
            (void)hipStreamDestroy(m->stream);
        }
    }
    if (e != hipSuccess) {
        delete m;
        return fail_hip(e, "hipStreamCreateWithFlags");
    }
    dev->mixers++;
    *out = m;
    return CGHIP_OK;
}

int cghip_mixer_destroy(cghip_mixer* mixer)
{
    if (!mixer)
        return CGHIP_OK;
    (void)hipSetDevice(mixer->dev->gpu);
    (void)hipStreamSynchronize(mixer->ops_stream);
    (void)hipStreamSynchronize(mixer->stream);
    (void)hipStreamSynchronize(mixer->readback_stream);
    (void)hipStreamDestroy(mixer->ops_stream);
    (void)hipStreamDestroy(mixer->stream);
    (void)hipStreamDestroy(mixer->readback_stream);
    mixer->dev->mixers--;
    delete mixer;
    return CGHIP_OK;
}

static void ticket_free(cghip_ticket* t, hipStream_t also_drain = nullptr)
{
    cghip_device* dev = t->dev;
    if (also_drain) { /* error path of render_async: copies and launches may already be queued on both streams, and
                         the events that normally order the release have not been recorded */
        (void)hipStreamSynchronize(also_drain);
        (void)hipStreamSynchronize(t->stream);
        if (t->readback_stream)
            (void)hipStreamSynchronize(t->readback_stream);
    }
    if (t->ev_done)
        (void)hipEventSynchronize(t->ev_done);
    else if (t->ev_stop)
        (void)hipEventSynchronize(t->ev_stop);
    /* all four are complete now (ev_ops and ev_start precede ev_stop on the stream): back to the pool */
    dev->event_put(cghip_device::EV_DONE, t->ev_done);
    dev->event_put(cghip_device::EV_TIMED, t->ev_start);
    dev->event_put(cghip_device::EV_TIMED, t->ev_stop);
    dev->event_put(cghip_device::EV_ORDER, t->ev_ops);
    dev->dev_free(t->d_out, t->out_bytes);
    dev->dev_free(t->d_ops, t->ops_bytes);
    for (int i = 0; i < 2; ++i)
        dev->dev_free(t->spill[i], t->spill_bytes[i]);
    for (void* kb : t->key_bufs)
        dev->dev_free(kb, t->key_buf_bytes);
    dev->pinned_free(t->h_ops, t->ops_bytes);
    dev->pinned_free(t->h_out, t->out_bytes);
    for (cghip_plane* p : t->planes)
        cghip_plane_release(p);
    dev->tickets_live--;
    delete t;
}

int cghip_mixer_render_async(cghip_mixer* mixer, const cghip_format* format, const cghip_surface* surfaces,
                             int nsurfaces, const cghip_draw* draws, int ndraws, unsigned flags, cghip_ticket** out)
{
    if (!mixer || !out)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    *out = nullptr;
    frame_plan plan;
    int        rc = plan_frame(format, surfaces, nsurfaces, draws, ndraws, &plan);
    if (rc != CGHIP_OK)
        return rc;
    for (cghip_plane* p : plan.planes)
        if (p->stub || p->dev != mixer->dev)
            return fail(CGHIP_ERR_INVALID, "plane lives on another device: route it with cghip_plane_peer_copy_async");

    cghip_device* dev = mixer->dev;
    CGHIP_TRY_HIP(hipSetDevice(dev->gpu));
    /* Direct readback: a frame that is ONE launch over the whole frame and is read back anyway writes its pixels straight
       into the pinned host buffer. The runtime's D2H copy into pinned memory is a shader copy on this platform
       (`__amd_rocclr_copyBuffer` in the kernel trace, no SDMA transfer): a second pass over the frame whose waves compete
       with the next frame's launch for wave slots, and how that went depended on the box (one channel, config 3 at 8
       bits: 0.69 - 0.93 ms per frame; the copy alone 0.59). Here the launch's own stores cross PCIe: 0.68 ms (87 % of the
       copy's rate) and 1.28 ms at 16 bits (92 %) on every box, no frame buffer in HBM, no second stream to order
       (tests/gpu_readback_direct.sh). Several channels on a GPU reach the link's rate either way (52 GB/s). The price: the
       launch's hipEvent time then includes the drain over PCIe, and cghip_ticket_device_ptr hands out the pinned
       buffer. Frames cut into several launches (RGBA0 travels in the frame buffer between them) and field-row
       readbacks keep the copy. CGHIP_READBACK_DIRECT=0 switches it off. */
    static const bool direct_env = [] { const char* v = std::getenv("CGHIP_READBACK_DIRECT"); return !v || std::atoi(v) != 0; }();
    const bool direct = direct_env && (flags & CGHIP_RENDER_READBACK) && !(flags & CGHIP_RENDER_FIELD_ROWS_ONLY) &&
                        plan.segments.size() == 1 && plan.segments[0].gx0 == 0 && plan.segments[0].gy0 == 0 &&
                        plan.segments[0].gx1 >= format->width && plan.segments[0].gy1 >= format->height;
    cghip_ticket* t = new cghip_ticket();
    t->dev = dev, t->stream = mixer->stream;
    if (direct && (mixer->direct_frames++ & 1)) /* consecutive direct frames alternate between two streams: while one
                                                   launch drains its stores over PCIe the next one composites */
        t->stream = mixer->readback_stream;
    dev->tickets_live++;
    t->out_bytes = (size_t)format->width * format->height * 4 * (format->bits / 8);
    const size_t ops_only = plan.ops.size() * sizeof(KOp), mask_bytes = plan.masks.size() * sizeof(uint64_t);
    t->ops_bytes = std::max<size_t>(ops_only + mask_bytes, 1); /* the op list, then the segments' op masks */
#define T_TRY(expr)                                                                                                    \
    do {                                                                                                               \
        int rc__ = (expr);                                                                                             \
        if (rc__ != CGHIP_OK) {                                                                                        \
            ticket_free(t, mixer->ops_stream);                                                                         \
            return rc__;                                                                                               \
        }                                                                                                              \
    } while (0)
#define T_HIP(expr)                                                                                                    \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess) {                                                                                       \
            ticket_free(t, mixer->ops_stream);                                                                         \
            return fail_hip(e__, #expr);                                                                               \
        }                                                                                                              \
    } while (0)
    if (direct)
        T_TRY(dev->pinned_alloc(t->out_bytes, &t->h_out));
    else
        T_TRY(dev->dev_alloc(t->out_bytes, &t->d_out));
    T_TRY(dev->dev_alloc(t->ops_bytes, &t->d_ops));
    T_TRY(dev->pinned_alloc(t->ops_bytes, &t->h_ops));
    if (!(flags & CGHIP_RENDER_UNTIMED)) /* one queue packet and one runtime call less per frame */
        T_TRY(dev->event_get(cghip_device::EV_TIMED, &t->ev_start));
    T_TRY(dev->event_get(cghip_device::EV_TIMED, &t->ev_stop));
    if (flags & CGHIP_RENDER_READBACK) /* without a readback the frame is done when its last launch is: ev_stop */
        T_TRY(dev->event_get(cghip_device::EV_DONE, &t->ev_done));

    /* The op list (a few KB) goes up from pinned memory. A channel that has its GPU to itself sends it on a stream of
       its own, so that it travels while the previous frame is still compositing (on the compositor's stream the copy
       sits between two launches: +2 us a frame on BASELINE config 3). With several channels on the GPU the queue never
       runs dry, and what counts is the number of runtime calls a frame costs its thread, all of which meet in the
       runtime's queue locks: the copy goes on the compositor's stream and the event record + stream wait that ordered
       it are not made (32 native channel threads, config 4: 26 100 -> 40 300 frames/s). The same holds for one channel
       whose frames are cheaper on the GPU than those three calls are on its thread (config 4, one channel: 25 100 ->
       28 000 frames/s), hence the work estimate. Reading the list in place from pinned memory was measured too:
       config 3's launch 0.213 -> 0.226 ms, 32 channels 36 600 frames/s. */
    const bool own_stream = dev->mixers.load() == 1 &&
                            (double)format->width * format->height * (double)plan.ops.size() >= 32e6;
    if (ops_only + mask_bytes > 0) {
        if (ops_only)
            std::memcpy(t->h_ops, plan.ops.data(), ops_only);
        if (mask_bytes)
            std::memcpy((char*)t->h_ops + ops_only, plan.masks.data(), mask_bytes);
        if (!own_stream) {
            T_HIP(hipMemcpyAsync(t->d_ops, t->h_ops, ops_only + mask_bytes, hipMemcpyHostToDevice, t->stream));
        } else {
            T_TRY(dev->event_get(cghip_device::EV_ORDER, &t->ev_ops));
            T_HIP(hipMemcpyAsync(t->d_ops, t->h_ops, ops_only + mask_bytes, hipMemcpyHostToDevice, mixer->ops_stream));
            T_HIP(hipEventRecord(t->ev_ops, mixer->ops_stream));
        }
    }
    const uint64_t* d_masks = (const uint64_t*)((const char*)t->d_ops + ops_only);
    /* the frame keeps its source planes alive and waits for their uploads (device.cpp:242-262 futures). A still or a
       routed frame is visited tick after tick: once its upload has been seen complete there is nothing to wait for,
       and a 16-layer 4:2:2 frame no longer queues 48 event waits in front of its launch. */
    for (cghip_plane* p : plan.planes) {
        cghip_plane_retain(p);
        t->planes.push_back(p);
        cghip_plane* root = p->parent ? p->parent : p; /* a view shares its parent's event */
        if (root->ready_seen.load(std::memory_order_acquire))
            continue;
        const hipError_t q = hipEventQuery(root->ready);
        if (q == hipSuccess) {
            root->ready_seen.store(true, std::memory_order_release);
            continue;
        }
        if (q != hipErrorNotReady)
            T_HIP(q);
        (void)hipGetLastError();
        T_HIP(hipStreamWaitEvent(t->stream, root->ready, 0));
    }
    if (t->ev_ops)
        T_HIP(hipStreamWaitEvent(t->stream, t->ev_ops, 0));
    if (t->ev_start)
        T_HIP(hipEventRecord(t->ev_start, t->stream));
    /* one launch per segment of the plan, in order on the mixer's stream; surfaces that are live across a cut travel
       through the frame buffer (RGBA0) and the ticket's spill buffers. An empty scene is one launch with zero ops
       that writes the all-zero frame (image_mixer.cpp:89-92). */
    for (int i = 0; i < 2; ++i)
        if (plan.spill_mask & (1u << (KS_RGBA1 + i))) {
            t->spill_bytes[i] = (size_t)format->width * format->height * 4 * (format->bits / 8);
            T_TRY(dev->dev_alloc(t->spill_bytes[i], &t->spill[i]));
        }
    t->key_buf_bytes = (size_t)format->width * format->height * (format->bits / 8);
    for (int i = 0; i < plan.n_key_bufs; ++i) {
        void* kb = nullptr;
        T_TRY(dev->dev_alloc(t->key_buf_bytes, &kb));
        t->key_bufs.push_back(kb);
    }
    auto key_buf = [&](int id) -> void* { return id >= 0 && id < (int)t->key_bufs.size() ? t->key_bufs[(size_t)id] : nullptr; };
    for (const segment& sg : plan.segments) {
        KSpill sp;
        sp.rgba1 = t->spill[0], sp.rgba2 = t->spill[1];
        sp.key0 = key_buf(sg.key_ld[0]), sp.key1 = key_buf(sg.key_ld[1]);
        sp.key0_st = key_buf(sg.key_st[0]), sp.key1_st = key_buf(sg.key_st[1]);
        sp.load_mask = sg.load_mask, sp.store_mask = sg.store_mask;
        sp.gx0 = sg.gx0, sp.gy0 = sg.gy0;
        sp.masks = d_masks + sg.mask_off, sp.ncols = sg.ncols, sp.nrows = sg.nrows, sp.nchunks = sg.nchunks;
        sp.col_shift = sg.col_shift, sp.dense = sg.dense ? 1 : 0, sp.fused = sg.fused ? 1 : 0, sp.one_temp = sg.one_temp;
        sp.mono = sg.mono;
        sp.mix_pairs = sg.mix_pairs ? 1 : 0;
        sp.yuv_lut = sg.yuv_lut ? 1 : 0;
        sp.host_out = direct ? 1 : 0;
        T_HIP(cghip_launch_composite((const KOp*)t->d_ops + sg.first, sg.count, direct ? t->h_out : t->d_out, format->width, format->height,
                                     format->bits, format->field_mode, sg.full ? 1 : 0, sg.fast ? 1 + sg.ext : 0, &sp,
                                     sg.gx1 - sg.gx0, sg.gy1 - sg.gy0, t->stream));
        t->launches++;
    }
    T_HIP(hipEventRecord(t->ev_stop, t->stream));
    if (direct) {
        T_HIP(hipEventRecord(t->ev_done, t->stream)); /* the launch's completion makes its stores to host memory visible */
    } else if (flags & CGHIP_RENDER_READBACK) {
        /* The D2H rides the channel's readback stream, ordered after the frame's last launch by ev_stop: the
           compositor's stream is free for the next frame while this one crosses PCIe (the reference polls a fence every
           2 ms on its one GL thread, device.cpp:264-302). d_out and h_out belong to this ticket alone, and ev_done, which
           releases them, is recorded behind the copy - no other hazard. */
        T_TRY(dev->pinned_alloc(t->out_bytes, &t->h_out));
        hipStream_t rs = mixer->readback_stream;
        t->readback_stream = rs;
        T_HIP(hipStreamWaitEvent(rs, t->ev_stop, 0));
        const int fm = format->field_mode;
        if ((flags & CGHIP_RENDER_FIELD_ROWS_ONLY) && (fm == 1 || fm == 2) && format->height > 1) {
            /* only the rows the kernel drew: a strided copy of every second row */
            const size_t row   = (size_t)format->width * 4 * (format->bits / 8);
            const size_t first = fm == 1 ? 0 : 1;
            const size_t rows  = ((size_t)format->height - first + 1) / 2;
            T_HIP(hipMemcpy2DAsync((char*)t->h_out + first * row, 2 * row, (const char*)t->d_out + first * row, 2 * row,
                                   row, rows, hipMemcpyDeviceToHost, rs));
        } else {
            T_HIP(hipMemcpyAsync(t->h_out, t->d_out, t->out_bytes, hipMemcpyDeviceToHost, rs));
        }
        T_HIP(hipEventRecord(t->ev_done, rs));
    }
#undef T_TRY
#undef T_HIP
    dev->frames_rendered++;
    dev->kernel_launches += t->launches;
    *out = t;
    return CGHIP_OK;
}

/* The allocation-free variant (include/cghip.h). Op list and hand-over surfaces live in the GPU's static scratch
 * array: [ CGHIP_NOALLOC_MAX_OPS KOps | RGBA1 | RGBA2 | KEY0 | KEY1 ], each surface region 256-byte aligned and only
 * present when the plan needs it. */
static std::mutex g_noalloc_mu[64];

int cghip_mixer_render_noalloc(cghip_mixer* mixer, const cghip_format* format, const cghip_surface* surfaces,
                               int nsurfaces, const cghip_draw* draws, int ndraws, void* device_out,
                               size_t device_out_bytes)
{
    if (!mixer || !device_out)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    frame_plan plan; /* host memory only */
    int        rc = plan_frame(format, surfaces, nsurfaces, draws, ndraws, &plan);
    if (rc != CGHIP_OK)
        return rc;
    cghip_device* dev = mixer->dev;
    for (cghip_plane* p : plan.planes)
        if (p->stub || p->dev != dev)
            return fail(CGHIP_ERR_INVALID, "plane lives on another device: route it with cghip_plane_peer_copy_async");
    const size_t out_bytes = (size_t)format->width * format->height * 4 * (format->bits / 8);
    if (device_out_bytes < out_bytes)
        return fail(CGHIP_ERR_INVALID, "device_out is smaller than width * height * 4 * (bits / 8)");
    if (((uintptr_t)device_out & 15) != 0)
        return fail(CGHIP_ERR_INVALID, "device_out must be 16-byte aligned");
    if (plan.ops.size() > CGHIP_NOALLOC_MAX_OPS)
        return fail(CGHIP_ERR_UNSUPPORTED, "frame has more than CGHIP_NOALLOC_MAX_OPS (256) ops: use cghip_mixer_render_async");
    const size_t mask_region = NOALLOC_MASK_REGION; /* the launches' op masks, behind the op list */
    const size_t ops_bytes = plan.ops.size() * sizeof(KOp), masks_bytes = plan.masks.size() * sizeof(uint64_t);
    if (masks_bytes > mask_region)
        return fail(CGHIP_ERR_UNSUPPORTED, "frame is too large or cut too often for cghip_mixer_render_noalloc's op-mask region");
    if (dev->gpu < 0 || dev->gpu >= 64)
        return fail(CGHIP_ERR_UNSUPPORTED, "gpu index beyond the scratch table");
    if (!dev->noalloc_staging || dev->noalloc_staging_bytes < noalloc_ops_bytes() + mask_region)
        return fail(CGHIP_ERR_INVALID, "device has no staging block for cghip_mixer_render_noalloc");

    CGHIP_TRY_HIP(hipSetDevice(dev->gpu));
    void*  base = nullptr;
    size_t cap  = 0;
    CGHIP_TRY_HIP(cghip_scratch_address(dev->gpu, &base, &cap));
    /* lay the hand-over surfaces out behind the op list; refuse what does not fit */
    const size_t ops_only   = noalloc_ops_bytes();
    const size_t ops_region = ops_only + mask_region;
    size_t       off        = ops_region;
    std::vector<void*> spill((size_t)(2 + plan.n_key_bufs), nullptr); /* RGBA1, RGBA2, then the key buffers (host memory) */
    for (int i = 0; i < 2 + plan.n_key_bufs; ++i)
        if (i >= 2 || (plan.spill_mask & (1u << (KS_RGBA1 + i)))) {
            const size_t need = round_up((size_t)format->width * format->height * (i < 2 ? 4 : 1) * (format->bits / 8), 256);
            if (off + need > cap) {
                char msg[200];
                snprintf(msg, sizeof msg,
                         "frame is cut into %zu launches whose hand-over surfaces need more than the %zu scratch bytes "
                         "of cghip_mixer_render_noalloc: use cghip_mixer_render_async",
                         plan.segments.size(), cap - ops_region);
                return fail(CGHIP_ERR_UNSUPPORTED, msg);
            }
            spill[(size_t)i] = (char*)base + off;
            off += need;
        }
    auto key_buf = [&](int id) -> void* { return id >= 0 && id < plan.n_key_bufs ? spill[(size_t)(2 + id)] : nullptr; };

    std::lock_guard<std::mutex> lock(g_noalloc_mu[dev->gpu]); /* one frame at a time in a GPU's scratch array */
    hipStream_t                 st = mixer->stream;
    /* From here on work may be queued on `st` that reads the staging block and the scratch array: EVERY way out goes
       through the stream synchronisation at the end, so that neither is handed to the next caller (nor the staging
       block rewritten) while launches or copies of this frame, failed or not, are still in flight. */
    hipError_t  e    = hipSuccess;
    const char* what = "";
#define NA_HIP(expr)                                                                                                   \
    do {                                                                                                               \
        if (e == hipSuccess) {                                                                                         \
            e = (expr);                                                                                                \
            if (e != hipSuccess)                                                                                       \
                what = #expr;                                                                                          \
        }                                                                                                              \
    } while (0)
    for (cghip_plane* p : plan.planes) { /* uploads still in flight */
        cghip_plane* root = p->parent ? p->parent : p;
        if (!root->ready_seen.load(std::memory_order_acquire))
            NA_HIP(hipStreamWaitEvent(st, root->ready, 0));
    }
    /* op list and masks go up from the device's pinned staging block (set up once, cghip_device_create) */
    char* stage = (char*)dev->noalloc_staging;
    if (ops_bytes) {
        std::memcpy(stage, plan.ops.data(), ops_bytes);
        NA_HIP(hipMemcpyAsync(base, stage, ops_bytes, hipMemcpyHostToDevice, st));
    }
    const uint64_t* d_masks = (const uint64_t*)((const char*)base + ops_only);
    if (masks_bytes) {
        std::memcpy(stage + ops_only, plan.masks.data(), masks_bytes);
        NA_HIP(hipMemcpyAsync((void*)d_masks, stage + ops_only, masks_bytes, hipMemcpyHostToDevice, st));
    }
    int launches = 0;
    for (const segment& sg : plan.segments) {
        if (e != hipSuccess)
            break;
        KSpill sp;
        sp.rgba1 = spill[0], sp.rgba2 = spill[1];
        sp.key0 = key_buf(sg.key_ld[0]), sp.key1 = key_buf(sg.key_ld[1]);
        sp.key0_st = key_buf(sg.key_st[0]), sp.key1_st = key_buf(sg.key_st[1]);
        sp.load_mask = sg.load_mask, sp.store_mask = sg.store_mask;
        sp.gx0 = sg.gx0, sp.gy0 = sg.gy0;
        sp.masks = d_masks + sg.mask_off, sp.ncols = sg.ncols, sp.nrows = sg.nrows, sp.nchunks = sg.nchunks;
        sp.col_shift = sg.col_shift, sp.dense = sg.dense ? 1 : 0, sp.fused = sg.fused ? 1 : 0, sp.one_temp = sg.one_temp;
        sp.mono = sg.mono;
        sp.mix_pairs = sg.mix_pairs ? 1 : 0;
        sp.yuv_lut = sg.yuv_lut ? 1 : 0;
        sp.host_out = 0; /* the caller's device memory */
        NA_HIP(cghip_launch_composite((const KOp*)base + sg.first, sg.count, device_out, format->width, format->height,
                                      format->bits, format->field_mode, sg.full ? 1 : 0, sg.fast ? 1 + sg.ext : 0, &sp,
                                      sg.gx1 - sg.gx0, sg.gy1 - sg.gy0, st));
        if (e == hipSuccess)
            ++launches;
    }
#undef NA_HIP
    /* the one way out: staging block and scratch array stay untouched until everything queued above is done */
    const hipError_t es = hipStreamSynchronize(st);
    if (e != hipSuccess)
        return fail_hip(e, what);
    if (es != hipSuccess)
        return fail_hip(es, "hipStreamSynchronize");
    dev->frames_rendered++;
    dev->kernel_launches += launches;
    return CGHIP_OK;
}

int cghip_ticket_wait(cghip_ticket* ticket, const uint8_t** data, size_t* size)
{
    if (!ticket)
        return fail(CGHIP_ERR_INVALID, "ticket is null");
    CGHIP_TRY_HIP(hipEventSynchronize(ticket->ev_done ? ticket->ev_done : ticket->ev_stop));
    if (data) {
        if (!ticket->h_out)
            return fail(CGHIP_ERR_INVALID, "frame was rendered without CGHIP_RENDER_READBACK");
        *data = (const uint8_t*)ticket->h_out;
    }
    if (size)
        *size = ticket->out_bytes;
    return CGHIP_OK;
}

int cghip_ticket_device_ptr(cghip_ticket* ticket, const void** device_ptr, size_t* size)
{
    if (!ticket || !device_ptr)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    *device_ptr = ticket->d_out ? ticket->d_out : ticket->h_out; /* direct readback: the pinned buffer, device-accessible */
    if (size)
        *size = ticket->out_bytes;
    return CGHIP_OK;
}

int cghip_ticket_kernel_ms(cghip_ticket* ticket, float* ms, int* launches)
{
    if (!ticket || !ms)
        return fail(CGHIP_ERR_INVALID, "bad arguments");
    CGHIP_TRY_HIP(hipEventSynchronize(ticket->ev_stop));
    if (!ticket->ev_start) { /* CGHIP_RENDER_UNTIMED: "not measured" is not "took 0 ms" */
        *ms = -1.0f;
        if (launches)
            *launches = ticket->launches;
        return CGHIP_OK;
    }
    CGHIP_TRY_HIP(hipEventElapsedTime(ms, ticket->ev_start, ticket->ev_stop));
    if (launches)
        *launches = ticket->launches;
    return CGHIP_OK;
}

int cghip_ticket_release(cghip_ticket* ticket)
{
    if (!ticket)
        return CGHIP_OK;
    ticket_free(ticket);
    return CGHIP_OK;
}

} /* extern "C" */
