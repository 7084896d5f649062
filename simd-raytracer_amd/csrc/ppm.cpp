// ASCII P3 writer with the exact byte layout of io/image/ppm.hpp:7-25:
// "P3\n<w> <h>\n255\n", then per pixel "<r> <g> <b>\t", a '\n' per row,
// channel = uint8(255.999 * clamp(c, 0, 1)) with the product taken in double.
#include <cstdio>

#include "rtk_internal.hpp"

namespace rtk {

// the same text from already quantised channels (rtk_frame_to_rgb8_device)
std::string format_ppm_rgb8(const uint8_t *rgb8, int width, int height) {
    std::string out;
    out.reserve(size_t(width) * size_t(height) * 12 + 32);
    char tmp[48];
    out.append(tmp, size_t(std::snprintf(tmp, sizeof(tmp), "P3\n%d %d\n255\n", width, height)));
    for (int y = 0; y < height; ++y) {
        const uint8_t *row = rgb8 + size_t(y) * size_t(width) * 3;
        for (int x = 0; x < width; ++x) {
            const int n = std::snprintf(tmp, sizeof(tmp), "%u %u %u\t", unsigned(row[x * 3]), unsigned(row[x * 3 + 1]), unsigned(row[x * 3 + 2]));
            out.append(tmp, size_t(n));
        }
        out.push_back('\n');
    }
    return out;
}

std::string format_ppm(const float *rgb, int width, int height) {
    std::string out;
    out.reserve(size_t(width) * size_t(height) * 12 + 32);
    char tmp[48];
    out.append(tmp, size_t(std::snprintf(tmp, sizeof(tmp), "P3\n%d %d\n255\n", width, height)));
    auto channel = [](float c) -> unsigned {
        const float k = (c < 0.0f) ? 0.0f : ((1.0f < c) ? 1.0f : c);   // std::clamp(c, 0.f, 1.f)
        return static_cast<unsigned>(static_cast<uint8_t>(255.999 * static_cast<double>(k)));
    };
    for (int y = 0; y < height; ++y) {
        const float *row = rgb + size_t(y) * size_t(width) * 3;
        for (int x = 0; x < width; ++x) {
            const int n = std::snprintf(tmp, sizeof(tmp), "%u %u %u\t", channel(row[x * 3]), channel(row[x * 3 + 1]),
                                        channel(row[x * 3 + 2]));
            out.append(tmp, size_t(n));
        }
        out.push_back('\n');
    }
    return out;
}

}  // namespace rtk
