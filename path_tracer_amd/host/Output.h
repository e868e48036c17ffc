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

}   // namespace dp
#endif
