/*
 * route_driver.cpp -- BASELINE configs[3] and configs[4] in the PRODUCT's topology: ONE process, N GPUs, one native
 * thread per channel, through libcghost.so -> libcghip.so. That is how the reference runs (one accelerator, one device
 * shared by every channel: accelerator/accelerator.cpp:27-34; one thread per video_channel: core/video_channel.cpp:
 * 125-193); bench.py's `--gpus N` under torch.distributed.run is one process PER GPU, which cannot route a frame from
 * one GPU's channel to another's.
 *
 *   route_driver --workload config5|config4 [--gpus N] [--channels C] [--frames F] [--width W --height H]
 *                [--route-ring 0|1] [--stand-in] [--readback] [--untimed]          ->  one JSON line
 *
 *   config5: every channel composites the config-3 8-bit stack (16 ycbcr 4:2:2 layers: key / keyed fill / chroma key /
 *            levels, 2160p). --route-ring 1 (default): layer 0 of a channel on GPU g is a ROUTE (core/producer/route/
 *            route_producer.cpp:40-126) of a frame that a producer on GPU (g + 1) mod N makes anew every tick, so every
 *            channel moves one 16.6 MB 4:2:2 layer over xGMI per frame (cghip_plane_peer_copy_async inside visit()).
 *   config4: every channel a 1080p MIX transition over a background (three BGRA8 sources), no routes.
 *   channel c runs on GPU c mod N (accelerator::create_image_mixer's placement).
 *   --stand-in: on a box with fewer than N GPUs the missing ones are PRIVATE device objects on GPU 0: to the host code
 *            and the C ABI a foreign device (own pools, own streams, hipMemcpyPeerAsync), so the whole path runs.
 * Reports aggregate frames/s, per-GPU frames/s and the peer-copy rate from the devices' own counters.
 */
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cghip.h"

extern "C" {
struct cghost_plane {
    int32_t     width, height, stride, bits;
    const void* data;
};
const char* cghost_last_error(void);
int         cghost_device_count(void);
int         cghost_mixer_create(int channel_id, int gpu_index, void** out);
int         cghost_mixer_create_on(void* other, int channel_id, void** out);
int         cghost_mixer_destroy(void* h);
int         cghost_mixer_set_kernel_timing(void* h, int on);
int         cghost_mixer_push(void* h, const cghip_transform* t);
int         cghost_frame_create(void* h, int pixel_format, int nplanes, const cghost_plane* planes, const cghip_coord* coords,
                                int ncoords, void** out);
int         cghost_frame_destroy(void* frame);
int         cghost_mixer_visit_frame(void* h, void* frame);
int         cghost_mixer_pop(void* h);
int         cghost_mixer_render_async(void* h, const cghip_format* f, int readback, void** out);
int         cghost_future_get(void* pending, void* out, size_t capacity, float* kernel_ms);
int         cghost_mixer_device_info(void* h, char* json, size_t capacity);
}

static void ck(int rc, const char* what)
{
    if (rc != 0) {
        std::fprintf(stderr, "%s failed: %s\n", what, cghost_last_error());
        std::exit(2);
    }
}
#define CK(expr) ck((expr), #expr)

static cghip_transform identity()
{
    cghip_transform t;
    std::memset(&t, 0, sizeof t);
    t.opacity = t.contrast = t.brightness = t.saturation = 1.0;
    t.fill_scale[0] = t.fill_scale[1] = t.clip_scale[0] = t.clip_scale[1] = 1.0;
    t.crop_lr[0] = t.crop_lr[1] = 1.0;
    t.pers_ur[0] = t.pers_lr[0] = t.pers_lr[1] = t.pers_ll[1] = 1.0;
    t.levels_max_input = t.levels_gamma = t.levels_max_output = 1.0;
    t.chroma_spill_suppress_saturation = 1.0;
    return t;
}

/* the 16 layer transforms of BASELINE configs[2] (tests/scenes.py: config3) */
static cghip_transform config3_layer(int i)
{
    cghip_transform t = identity();
    t.layer_depth     = 1;
    switch (i % 4) {
        case 0: t.is_key = 1; break;
        case 1: t.opacity = 0.8; break;
        case 2:
            t.opacity = 0.75, t.chroma_enable = 1, t.chroma_target_hue = 120.0, t.chroma_hue_width = 0.1;
            t.chroma_min_saturation = 0.2, t.chroma_min_brightness = 0.2, t.chroma_softness = 0.1;
            t.chroma_spill_suppress = 36.0, t.chroma_spill_suppress_saturation = 0.5;
            break;
        default:
            t.opacity = 0.7, t.levels_min_input = 0.06, t.levels_max_input = 0.92, t.levels_gamma = 1.2;
            break;
    }
    return t;
}

static long long json_int(const std::string& s, const char* key)
{
    const std::string k = std::string("\"") + key + "\":";
    const size_t      p = s.find(k);
    return p == std::string::npos ? -1 : std::atoll(s.c_str() + p + k.size());
}

struct args_t {
    std::string workload = "config5";
    int         gpus = 0, channels = 0, frames = 100, W = 0, H = 0, route_ring = 1, stand_in = 0, readback = 0, timed = 1;
};

int main(int argc, char** argv)
{
    args_t a;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto              val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (k == "--workload") a.workload = val();
        else if (k == "--gpus") a.gpus = std::atoi(val());
        else if (k == "--channels") a.channels = std::atoi(val());
        else if (k == "--frames") a.frames = std::atoi(val());
        else if (k == "--width") a.W = std::atoi(val());
        else if (k == "--height") a.H = std::atoi(val());
        else if (k == "--route-ring") a.route_ring = std::atoi(val());
        else if (k == "--stand-in") a.stand_in = 1;
        else if (k == "--readback") a.readback = 1;
        else if (k == "--untimed") a.timed = 0;
        else {
            std::fprintf(stderr, "unknown argument %s\n", k.c_str());
            return 2;
        }
    }
    const bool c5 = a.workload == "config5";
    if (!c5 && a.workload != "config4") {
        std::fprintf(stderr, "--workload config5 | config4\n");
        return 2;
    }
    const int ndev = cghost_device_count();
    if (ndev < 1) {
        std::fprintf(stderr, "no HIP device\n");
        return 2;
    }
    int N = a.gpus > 0 ? a.gpus : ndev;
    if (N > ndev && !a.stand_in) {
        std::fprintf(stderr, "%d GPUs asked for, %d visible (use --stand-in to rehearse with private device objects)\n", N, ndev);
        return 2;
    }
    const int W = a.W ? a.W : (c5 ? 3840 : 1920), H = a.H ? a.H : (c5 ? 2160 : 1080);
    const int channels = a.channels ? a.channels : (c5 ? 8 * N : 32);
    const int nlayers = c5 ? 16 : 3;
    /* GPU g of the job -> gpu_index for cghost_mixer_create: a real GPU, or a private device object on GPU 0 */
    auto gpu_index = [&](int g) { return g < ndev ? g : -100; };
    int  stand_ins = 0;
    for (int g = 0; g < N; ++g)
        stand_ins += g >= ndev;

    /* sources: smooth planes with a little noise; every GPU holds ONE set that its channels share (a still shown by
       several channels, like a logo or a looping clip) */
    std::vector<std::vector<uint8_t>> planes;
    uint32_t                          x = 2463534242u;
    auto fill = [&](size_t n, int lo, int hi) {
        std::vector<uint8_t> b(n);
        for (size_t i = 0; i < n; ++i) {
            x ^= x << 13, x ^= x >> 17, x ^= x << 5;
            b[i] = (uint8_t)(lo + (int)((i * 7 / 64 + (x >> 27)) % (unsigned)(hi - lo)));
        }
        return b;
    };
    if (c5) {
        planes.push_back(fill((size_t)W * H, 16, 235));
        planes.push_back(fill((size_t)(W / 2) * H, 16, 240));
        planes.push_back(fill((size_t)(W / 2) * H, 16, 240));
    } else {
        for (int k = 0; k < 3; ++k) {
            std::vector<uint8_t> b((size_t)W * H * 4);
            for (size_t i = 0; i < b.size(); i += 4) {
                x ^= x << 13, x ^= x >> 17, x ^= x << 5;
                const uint32_t al = x >> 24;
                b[i] = (uint8_t)(((x & 255) * al) / 255), b[i + 1] = (uint8_t)((((x >> 8) & 255) * al) / 255);
                b[i + 2] = (uint8_t)((((x >> 16) & 255) * al) / 255), b[i + 3] = (uint8_t)al;
            }
            planes.push_back(std::move(b));
        }
    }
    auto make_frame = [&](void* factory, int k) {
        void* f = nullptr;
        if (c5) {
            cghost_plane p[3] = {{W, H, 1, 8, planes[0].data()}, {W / 2, H, 1, 8, planes[1].data()}, {W / 2, H, 1, 8, planes[2].data()}};
            CK(cghost_frame_create(factory, CGHIP_PF_YCBCR, 3, p, nullptr, 0, &f));
        } else {
            cghost_plane p{W, H, 4, 8, planes[(size_t)k % planes.size()].data()};
            CK(cghost_frame_create(factory, CGHIP_PF_BGRA, 1, &p, nullptr, 0, &f));
        }
        return f;
    };

    /* per GPU: a factory mixer (the producers' frame_factory of that GPU) and the shared stills */
    std::vector<void*>              factory((size_t)N, nullptr);
    std::vector<std::vector<void*>> stills((size_t)N);
    for (int g = 0; g < N; ++g) {
        CK(cghost_mixer_create(1000 + g, gpu_index(g), &factory[(size_t)g]));
        for (int k = 0; k < nlayers; ++k)
            stills[(size_t)g].push_back(make_frame(factory[(size_t)g], k));
    }
    auto info_of = [&](int g) {
        std::string s(4096, '\0');
        CK(cghost_mixer_device_info(factory[(size_t)g], &s[0], s.size()));
        s.resize(std::strlen(s.c_str()));
        return s;
    };

    std::mutex               mu;
    std::condition_variable  cv;
    int                      ready = 0;
    bool                     go    = false;
    std::atomic<int>         done{0};
    std::vector<std::thread> threads;
    std::chrono::steady_clock::time_point t0, t1;
    std::vector<long long>   peer_bytes0((size_t)N, 0), peer_copies0((size_t)N, 0);
    const cghip_format       fmt{W, H, W, H, 8, 0};
    for (int c = 0; c < channels; ++c)
        threads.emplace_back([&, c] {
            const int g = c % N, src_gpu = (g + 1) % N;
            void*     m = nullptr;
            /* the channel's mixer, on the device object of its GPU (shared with that GPU's factory and other channels) */
            CK(cghost_mixer_create_on(factory[(size_t)g], c + 1, &m));
            CK(cghost_mixer_set_kernel_timing(m, a.timed));
            const bool routed = c5 && a.route_ring && N > 1;
            auto submit = [&](int tick) {
                void* fresh = nullptr;
                if (routed) /* the neighbour GPU's producer makes this tick's frame; visit() copies it over xGMI */
                    fresh = make_frame(factory[(size_t)src_gpu], 0);
                if (c5) {
                    for (int i = 0; i < 16; ++i) {
                        const cghip_transform t = config3_layer(i);
                        CK(cghost_mixer_push(m, &t));
                        CK(cghost_mixer_visit_frame(m, (i == 0 && fresh) ? fresh : stills[(size_t)g][(size_t)i]));
                        CK(cghost_mixer_pop(m));
                    }
                } else {
                    const double    delta = (tick % 25) / 25.0; /* transition_producer.cpp:159-163 */
                    cghip_transform bg = identity(), layer = identity(), ta = identity(), tb = identity();
                    bg.layer_depth = 1, layer.layer_depth = 1;
                    ta.is_mix = tb.is_mix = 1, ta.opacity = 1.0 - delta, tb.opacity = delta;
                    auto& st = stills[(size_t)g];
                    CK(cghost_mixer_push(m, &bg)), CK(cghost_mixer_visit_frame(m, st[0])), CK(cghost_mixer_pop(m));
                    CK(cghost_mixer_push(m, &layer));
                    CK(cghost_mixer_push(m, &ta)), CK(cghost_mixer_visit_frame(m, st[1])), CK(cghost_mixer_pop(m));
                    CK(cghost_mixer_push(m, &tb)), CK(cghost_mixer_visit_frame(m, st[2])), CK(cghost_mixer_pop(m));
                    CK(cghost_mixer_pop(m));
                }
                void* pending = nullptr;
                CK(cghost_mixer_render_async(m, &fmt, a.readback, &pending));
                if (fresh) /* the frame that shows it holds what it needs (planes, pinned buffers) until it is done */
                    cghost_frame_destroy(fresh);
                return pending;
            };
            auto run = [&](int n) { /* two frames in flight, like core::mixer (mixer.cpp:85-91) */
                void* prev = nullptr;
                for (int i = 0; i < n; ++i) {
                    void* cur = submit(i);
                    if (prev)
                        CK(cghost_future_get(prev, nullptr, 0, nullptr));
                    prev = cur;
                }
                if (prev)
                    CK(cghost_future_get(prev, nullptr, 0, nullptr));
            };
            run(c5 ? 10 : 50); /* warm-up: pools grow, clocks rise, peer access is asked for */
            {
                std::unique_lock<std::mutex> lk(mu);
                if (++ready == channels) {
                    for (int gg = 0; gg < N; ++gg) {
                        const std::string s = info_of(gg);
                        peer_bytes0[(size_t)gg] = json_int(s, "peer_copy_bytes"), peer_copies0[(size_t)gg] = json_int(s, "peer_copies");
                    }
                    go = true;
                    t0 = std::chrono::steady_clock::now();
                    cv.notify_all();
                } else {
                    cv.wait(lk, [&] { return go; });
                }
            }
            run(a.frames);
            if (++done == channels)
                t1 = std::chrono::steady_clock::now();
            cghost_mixer_destroy(m);
        });
    for (auto& t : threads)
        t.join();
    const double s = std::chrono::duration<double>(t1 - t0).count();
    /* peer copies are counted by the DESTINATION device */
    long long peer_bytes = 0, peer_copies = 0;
    std::string access = "{";
    for (int g = 0; g < N; ++g) {
        const std::string inf = info_of(g);
        peer_bytes += json_int(inf, "peer_copy_bytes") - peer_bytes0[(size_t)g];
        peer_copies += json_int(inf, "peer_copies") - peer_copies0[(size_t)g];
        const size_t p = inf.find("\"peer_access\":");
        if (g)
            access += ",";
        access += "\"" + std::to_string(g) + "\":" + (p == std::string::npos ? "{}" : inf.substr(p + 14, inf.size() - p - 15));
    }
    access += "}";
    std::printf("{\"workload\": \"%s %dx%d\", \"topology\": \"one process, %d GPU(s), one thread per channel\", \"gpus\": %d, "
                "\"gpus_visible\": %d, \"stand_in_devices\": %d, \"channels\": %d, \"route_ring\": %d, \"readback\": %d, "
                "\"kernel_timing\": %d, \"frames_per_channel\": %d, \"seconds\": %.4f, \"frames_per_s\": %.1f, "
                "\"per_gpu_fps\": %.1f, \"per_channel_fps\": %.1f, \"peer_copies\": %lld, \"peer_copy_bytes\": %lld, "
                "\"peer_copy_GBps\": %.3f, \"peer_access\": %s}\n",
                a.workload.c_str(), W, H, N, N, ndev, stand_ins, channels, (c5 && a.route_ring && N > 1) ? 1 : 0, a.readback,
                a.timed, a.frames, s, channels * (double)a.frames / s, channels * (double)a.frames / s / N, a.frames / s,
                peer_copies, peer_bytes, peer_bytes / s / 1e9, access.c_str());
    for (int g = 0; g < N; ++g) {
        for (void* f : stills[(size_t)g])
            cghost_frame_destroy(f);
        cghost_mixer_destroy(factory[(size_t)g]);
    }
    return 0;
}
