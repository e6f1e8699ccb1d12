/*
 * core_types.h -- the reference's core value types that cross the image-mixer interface, restated so the
 * host side of the HIP mixer builds without boost/TBB (neither is in this image). In the real tree the
 * adapter includes the reference headers instead; names, fields, defaults and
 * semantics here follow them one to one:
 *   core/frame/frame_transform.h:34-147   chroma, levels, corners, rectangle, image_transform, frame_transform
 *   core/frame/frame_transform.cpp:32-113 image_transform::operator*=
 *   core/frame/geometry.h:30-64, geometry.cpp:64-89   frame_geometry
 *   core/frame/pixel_format.h:28-76       pixel_format, pixel_format_desc
 *   core/mixer/image/blend_modes.h:28-61  blend_mode
 *   core/video_format.h / video_format.cpp:34-113     video_format_desc (fields the mixer reads)
 */
#pragma once

#include <array>
#include <cmath>
#include <string>
#include <vector>

namespace caspar { namespace core {

enum class blend_mode
{
    normal = 0, lighten, darken, multiply, average, add, subtract, difference, negation, exclusion, screen, overlay,
    soft_light, hard_light, color_dodge, color_burn, linear_dodge, linear_burn, linear_light, vivid_light, pin_light,
    hard_mix, reflect, glow, phoenix, contrast, saturation, color, luminosity, mix, blend_mode_count
};

struct chroma
{
    bool   enable                    = false;
    bool   show_mask                 = false;
    double target_hue                = 0.0;
    double hue_width                 = 0.0;
    double min_saturation            = 0.0;
    double min_brightness            = 0.0;
    double softness                  = 0.0;
    double spill_suppress            = 0.0;
    double spill_suppress_saturation = 1.0;
};

struct levels final
{
    double min_input  = 0.0;
    double max_input  = 1.0;
    double gamma      = 1.0;
    double min_output = 0.0;
    double max_output = 1.0;
};

struct corners final
{
    std::array<double, 2> ul = {0.0, 0.0};
    std::array<double, 2> ur = {1.0, 0.0};
    std::array<double, 2> lr = {1.0, 1.0};
    std::array<double, 2> ll = {0.0, 1.0};
};

struct rectangle final
{
    std::array<double, 2> ul = {0.0, 0.0};
    std::array<double, 2> lr = {1.0, 1.0};
};

struct image_transform final
{
    double opacity    = 1.0;
    double contrast   = 1.0;
    double brightness = 1.0;
    double saturation = 1.0;

    std::array<double, 2> anchor           = {0.0, 0.0};
    std::array<double, 2> fill_translation = {0.0, 0.0};
    std::array<double, 2> fill_scale       = {1.0, 1.0};
    std::array<double, 2> clip_translation = {0.0, 0.0};
    std::array<double, 2> clip_scale       = {1.0, 1.0};
    double                angle            = 0.0;
    rectangle             crop;
    corners               perspective;
    core::levels          levels;
    core::chroma          chroma;

    bool             is_key      = false;
    bool             invert      = false;
    bool             is_mix      = false;
    core::blend_mode blend_mode  = blend_mode::normal;
    int              layer_depth = 0;

    image_transform& operator*=(const image_transform& other);
    image_transform  operator*(const image_transform& other) const { return image_transform(*this) *= other; }
};

struct audio_transform final
{
    double volume = 1.0;
};

struct frame_transform final
{
    core::image_transform image_transform;
    core::audio_transform audio_transform;
};

enum class pixel_format
{
    gray = 0, bgra, rgba, argb, abgr, ycbcr, ycbcra, luma, bgr, rgb, uyvy, count, invalid,
};

struct pixel_format_desc final
{
    struct plane
    {
        int linesize = 0;
        int width    = 0;
        int height   = 0;
        int size     = 0;
        int stride   = 0; /* samples per texel */
        int bits     = 8; /* extension: 16 selects 2-byte samples */

        plane() = default;
        plane(int width, int height, int stride, int bits = 8)
            : linesize(width * stride * (bits / 8))
            , width(width)
            , height(height)
            , size(width * height * stride * (bits / 8))
            , stride(stride)
            , bits(bits)
        {
        }
    };

    pixel_format_desc() = default;
    pixel_format_desc(pixel_format format)
        : format(format)
    {
    }

    pixel_format       format = pixel_format::invalid;
    std::vector<plane> planes;
};

class frame_geometry
{
  public:
    enum class geometry_type
    {
        quad
    };

    struct coord
    {
        double vertex_x  = 0.0;
        double vertex_y  = 0.0;
        double texture_x = 0.0;
        double texture_y = 0.0;
        double texture_r = 0.0;
        double texture_q = 1.0;

        coord() = default;
        coord(double vx, double vy, double tx, double ty)
            : vertex_x(vx)
            , vertex_y(vy)
            , texture_x(tx)
            , texture_y(ty)
        {
        }
    };

    frame_geometry(geometry_type type, std::vector<coord> data)
        : type_(type)
        , data_(std::move(data))
    {
    }

    geometry_type             type() const { return type_; }
    const std::vector<coord>& data() const { return data_; }

    static const frame_geometry& get_default()
    {
        static const frame_geometry g(geometry_type::quad,
                                      {{0.0, 0.0, 0.0, 0.0}, {1.0, 0.0, 1.0, 0.0}, {1.0, 1.0, 1.0, 1.0}, {0.0, 1.0, 0.0, 1.0}});
        return g;
    }
    static const frame_geometry& get_default_vflip()
    {
        static const frame_geometry g(geometry_type::quad,
                                      {{0.0, 0.0, 0.0, 1.0}, {1.0, 0.0, 1.0, 1.0}, {1.0, 1.0, 1.0, 0.0}, {0.0, 1.0, 0.0, 0.0}});
        return g;
    }

  private:
    geometry_type      type_;
    std::vector<coord> data_;
};

struct video_format_desc final
{
    int          width         = 0;
    int          height        = 0;
    int          square_width  = 0;
    int          square_height = 0;
    int          field_count   = 1;
    double       fps           = 0.0;
    int          size          = 0; /* width*height*4 (video_format.cpp:108) */
    std::wstring name;
    /* extensions carried next to the reference fields */
    int bits       = 8;
    int field_mode = 0;

    video_format_desc() = default;
    video_format_desc(int width, int height, int square_width, int square_height, int field_count, double fps,
                      std::wstring name)
        : width(width)
        , height(height)
        , square_width(square_width)
        , square_height(square_height)
        , field_count(field_count)
        , fps(fps)
        , size(width * height * 4)
        , name(std::move(name))
    {
    }
};

}} // namespace caspar::core
