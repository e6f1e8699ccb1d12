/*
 * channels_driver.cpp -- N channels of BASELINE config 4 (1080p MIX transition over a background, 33 MB per frame) on
 * ONE GPU, each driven by its own native thread through libcghost.so -> libcghip.so, the way video_channel threads
 * drive their mixers in the reference (core/video_channel.cpp:125-193). bench.py drives channels from one Python
 * thread, which tops out near 25 000 submissions a second; this measures what the runtime itself sustains.
 *
 *   channels_driver <channels> <frames per channel> [width height [timed [churn]]]   ->  one JSON line
 *   timed = 0: frames without the hipEvent pair that feeds a diagnostics graph (CGHIP_RENDER_UNTIMED)
 *   churn = 1: lifetime stress. Every tick one source frame of every channel is created anew (pinned buffer, upload) and
 *              the one it replaces is destroyed right after the frame that used it was submitted, i.e. while its
 *              upload and the kernels reading it may still be in flight; the last frame of every channel is read back
 *              and must equal the first one byte for byte (the sources never change content)
 */
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
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

int main(int argc, char** argv)
{
    const int channels = argc > 1 ? std::atoi(argv[1]) : 32, frames = argc > 2 ? std::atoi(argv[2]) : 500;
    const int W = argc > 4 ? std::atoi(argv[3]) : 1920, H = argc > 4 ? std::atoi(argv[4]) : 1080;
    const int timed = argc > 5 ? std::atoi(argv[5]) : 1;
    const int churn = argc > 6 ? std::atoi(argv[6]) : 0;
    std::atomic<int> mismatches{0};
    if (cghost_device_count() < 1) {
        std::fprintf(stderr, "no HIP device\n");
        return 2;
    }
    /* premultiplied BGRA noise, one buffer per layer shared by all channels (every channel uploads its own copy) */
    std::vector<std::vector<uint8_t>> src(3, std::vector<uint8_t>((size_t)W * H * 4));
    uint32_t                          x = 12345;
    for (auto& b : src)
        for (size_t i = 0; i < b.size(); i += 4) {
            x ^= x << 13, x ^= x >> 17, x ^= x << 5;
            const uint32_t a = x >> 24;
            b[i] = (uint8_t)(((x & 255) * a) / 255), b[i + 1] = (uint8_t)((((x >> 8) & 255) * a) / 255);
            b[i + 2] = (uint8_t)((((x >> 16) & 255) * a) / 255), b[i + 3] = (uint8_t)a;
        }

    std::mutex              mu;
    std::condition_variable cv;
    int                     ready = 0;
    bool                    go    = false;
    std::atomic<int>        done{0};
    std::atomic<long long>  submit_ns{0}, wait_ns{0}; /* time the channel threads spend submitting / waiting */
    std::vector<std::thread> threads;
    std::chrono::steady_clock::time_point t0, t1;
    for (int c = 0; c < channels; ++c)
        threads.emplace_back([&, c] {
            void* m = nullptr;
            CK(cghost_mixer_create(c + 1, 0, &m));
            CK(cghost_mixer_set_kernel_timing(m, timed));
            void* fr[3];
            for (int k = 0; k < 3; ++k) {
                cghost_plane p{W, H, 4, 8, src[k].data()};
                CK(cghost_frame_create(m, CGHIP_PF_BGRA, 1, &p, nullptr, 0, &fr[k]));
            }
            const cghip_format fmt{W, H, W, H, 8, 0};
            void*              stale = nullptr;
            auto submit = [&](int tick, int readback = 0) {
                if (churn) { /* replace source tick % 3; the old frame goes right after this submission */
                    const int    k = tick % 3;
                    void*        fresh = nullptr;
                    cghost_plane p{W, H, 4, 8, src[k].data()};
                    CK(cghost_frame_create(m, CGHIP_PF_BGRA, 1, &p, nullptr, 0, &fresh));
                    std::swap(fresh, fr[k]);
                    stale = fresh;
                }
                const double    delta = churn ? 0.4 : (tick % 25) / 25.0; /* transition_producer.cpp:159-163 */
                cghip_transform bg = identity(), layer = identity(), a = identity(), b = identity();
                bg.layer_depth = 1, layer.layer_depth = 1;
                a.is_mix = b.is_mix = 1, a.opacity = 1.0 - delta, b.opacity = delta;
                CK(cghost_mixer_push(m, &bg)), CK(cghost_mixer_visit_frame(m, fr[0])), CK(cghost_mixer_pop(m));
                CK(cghost_mixer_push(m, &layer));
                CK(cghost_mixer_push(m, &a)), CK(cghost_mixer_visit_frame(m, fr[1])), CK(cghost_mixer_pop(m));
                CK(cghost_mixer_push(m, &b)), CK(cghost_mixer_visit_frame(m, fr[2])), CK(cghost_mixer_pop(m));
                CK(cghost_mixer_pop(m));
                void* pending = nullptr;
                CK(cghost_mixer_render_async(m, &fmt, readback, &pending));
                if (stale) {
                    cghost_frame_destroy(stale);
                    stale = nullptr;
                }
                return pending;
            };
            auto run = [&](int n) { /* two frames in flight, like core::mixer (mixer.cpp:85-91) */
                void* prev = nullptr;
                for (int i = 0; i < n; ++i) {
                    const auto a   = std::chrono::steady_clock::now();
                    void*      cur = submit(i);
                    const auto b   = std::chrono::steady_clock::now();
                    if (prev)
                        CK(cghost_future_get(prev, nullptr, 0, nullptr));
                    const auto e = std::chrono::steady_clock::now();
                    submit_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
                    wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(e - b).count();
                    prev = cur;
                }
                if (prev)
                    CK(cghost_future_get(prev, nullptr, 0, nullptr));
            };
            run(50); /* warm-up: pools grow, clocks rise */
            submit_ns = 0, wait_ns = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (++ready == channels) {
                    go = true;
                    t0 = std::chrono::steady_clock::now();
                    cv.notify_all();
                } else {
                    cv.wait(lk, [&] { return go; });
                }
            }
            std::vector<uint8_t> first, last;
            if (churn) {
                first.resize((size_t)W * H * 4), last.resize(first.size());
                CK(cghost_future_get(submit(0, 1), first.data(), first.size(), nullptr));
            }
            run(frames);
            if (++done == channels)
                t1 = std::chrono::steady_clock::now();
            if (churn) {
                CK(cghost_future_get(submit(frames, 1), last.data(), last.size(), nullptr));
                if (first != last)
                    ++mismatches;
            }
            for (int k = 0; k < 3; ++k)
                cghost_frame_destroy(fr[k]);
            cghost_mixer_destroy(m);
        });
    for (auto& t : threads)
        t.join();
    const double s = std::chrono::duration<double>(t1 - t0).count();
    std::printf("{\"workload\": \"config4 %dx%d\", \"channels\": %d, \"kernel_timing\": %d, \"frames_per_channel\": %d, \"seconds\": %.4f, "
                "\"frames_per_s\": %.1f, \"per_channel_fps\": %.1f, \"submit_us_per_frame\": %.1f, "
                "\"wait_us_per_frame\": %.1f}\n",
                W, H, channels, timed, frames, s, channels * (double)frames / s, frames / s,
                submit_ns / 1e3 / ((double)channels * frames), wait_ns / 1e3 / ((double)channels * frames));
    if (churn)
        std::printf("{\"churn\": 1, \"channels_with_a_changed_frame\": %d}\n", mismatches.load());
    return mismatches.load() ? 3 : 0;
}
