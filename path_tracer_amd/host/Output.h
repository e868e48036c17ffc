// Output.h — gamma, quantisation and the ASCII PPM (reference: writeColour, Path Tracer.cpp:62-74,
// and the writer at :316-322). O(pixels), runs once per frame, stays on the host.
#ifndef PT_HOST_OUTPUT_H
#define PT_HOST_OUTPUT_H

#include <algorithm>
#include <cmath>
#include <ostream>
#include <vector>

#include "PhysicsVector.h"

namespace dp {

inline int quantiseChannel(double sum, int samplesPerPixel) {
    const double scale = 1.0 / static_cast<double>(samplesPerPixel);
    const double v = std::sqrt(scale * sum);
    return static_cast<int>(256 * std::clamp(v, 0.0, 0.999));
}

inline void writeColour(std::ostream& out, const PhysicsVector<3>& c, int samplesPerPixel) {
    out << quantiseChannel(c.x(), samplesPerPixel) << ' ' << quantiseChannel(c.y(), samplesPerPixel) << ' '
        << quantiseChannel(c.z(), samplesPerPixel) << '\n';
}

// pixels: top row first, width*height sums over samplesPerPixel samples.
inline void writeImage(std::ostream& out, const std::vector<PhysicsVector<3>>& pixels, int width, int height, int samplesPerPixel) {
    out << "P3\n" << width << ' ' << height << "\n255\n";
    for (const auto& p : pixels) writeColour(out, p, samplesPerPixel);
}

// The same picture from bytes already quantised on the device (pt_quantise_rgb8): ASCII P3, byte for
// byte what writeImage produces, or binary P6 (three bytes per pixel instead of up to twelve characters).
inline void writeImageP3(std::ostream& out, const std::vector<unsigned char>& rgb8, int width, int height) {
    out << "P3\n" << width << ' ' << height << "\n255\n";
    for (std::size_t i = 0; i + 2 < rgb8.size(); i += 3)
        out << int(rgb8[i]) << ' ' << int(rgb8[i + 1]) << ' ' << int(rgb8[i + 2]) << '\n';
}
inline void writeImageP6(std::ostream& out, const std::vector<unsigned char>& rgb8, int width, int height) {
    out << "P6\n" << width << ' ' << height << "\n255\n";
    out.write(reinterpret_cast<const char*>(rgb8.data()), std::streamsize(rgb8.size()));
}

}   // namespace dp
#endif
