// SceneGenerator.h — the reference's scene (Path Tracer.cpp:182-260): five fixed spheres, then
// up to `count` random small spheres placed by rejection. The reference seeds its two engines
// from std::random_device (Path Tracer.cpp:49, VectorFunc.cpp:16); here the two seeds are
// arguments, and with the reference's count (250), half extent (10) and thresholds (0.6, 0.9)
// the doubles that come out are those the reference's code produces from the same seeds.
#ifndef PT_HOST_SCENEGENERATOR_H
#define PT_HOST_SCENEGENERATOR_H

#include <cstdint>
#include <memory>
#include <random>
#include <vector>

#include "Dielectric.h"
#include "HittableList.h"
#include "Lambertian.h"
#include "Metal.h"
#include "Sphere.h"
#include "VectorFunc.h"

namespace dp {

struct SceneOptions {
    int numberOfSpheres = 250;        // Path Tracer.cpp:199
    std::uint32_t seedA = 1;          // RNG-A: selector, radius, x, z, fuzz (Path Tracer.cpp:48-52)
    std::uint32_t seedB = 2;          // RNG-B: albedo colours (VectorFunc.cpp:15-19)
    double halfExtent = 10.0;         // Path Tracer.cpp:218
    double diffuseBelow = 0.6;        // Path Tracer.cpp:242
    double metalBelow = 0.9;          // Path Tracer.cpp:247
    bool useGrid = true;              // overlap test through a uniform grid: same decisions, same scene
};

// Uniform grid over the placement square for the overlap test of Path Tracer.cpp:220-226. The
// reference scans every object placed so far (O(n) per attempt, O(n^2) per scene); whether a
// candidate clips ANY object does not depend on the scan order, so asking only the cells a clipping
// sphere could lie in gives the same accept / reject decisions, consumes the same random numbers
// and yields the same scene bit for bit. Random spheres have r < 0.4, so two of them can only clip
// when their centres are within 0.8 in x and z: one ring of 0.8-wide cells. The five fixed spheres
// (radius 1 and 1000) are always tested directly.
class PlacementGrid {
    static constexpr double kCell = 0.8;
    double m_lo;
    int m_n;
    std::vector<std::vector<std::size_t>> m_cells;   // indices into the world list
    int cellOf(double v) const {
        int c = static_cast<int>((v - m_lo) / kCell);
        return c < 0 ? 0 : (c >= m_n ? m_n - 1 : c);
    }

public:
    explicit PlacementGrid(double halfExtent) : m_lo(-halfExtent), m_n(static_cast<int>(2 * halfExtent / kCell) + 1), m_cells(std::size_t(m_n) * m_n) {}
    void insert(const PhysicsVector<3>& c, std::size_t index) { m_cells[std::size_t(cellOf(c.z())) * m_n + cellOf(c.x())].push_back(index); }
    template <class F>
    bool anyNear(const PhysicsVector<3>& c, F&& clips) const {
        const int cx = cellOf(c.x()), cz = cellOf(c.z());
        for (int z = (cz > 0 ? cz - 1 : 0); z <= (cz + 1 < m_n ? cz + 1 : m_n - 1); ++z)
            for (int x = (cx > 0 ? cx - 1 : 0); x <= (cx + 1 < m_n ? cx + 1 : m_n - 1); ++x)
                for (std::size_t idx : m_cells[std::size_t(z) * m_n + x])
                    if (clips(idx)) return true;
        return false;
    }
};

using colour = PhysicsVector<3>;
using point3D = PhysicsVector<3>;

// Path Tracer.cpp:182-195, in list order.
inline void addFixedSpheres(HittableList& world) {
    auto ground = std::make_shared<Lambertian>(colour{0.5, 0.5, 0.5});
    auto redDiffuse = std::make_shared<Lambertian>(colour{0.9, 0.1, 0.1});
    auto greyMetal = std::make_shared<Metal>(colour{0.8, 0.8, 0.8}, 0);
    auto greyFuzzy = std::make_shared<Metal>(colour{0.8, 0.8, 0.8}, 0.8);
    auto glass = std::make_shared<Dielectric>(1.5);
    world.add(std::make_shared<Sphere>(point3D{2, 1, 4}, 1, redDiffuse));
    world.add(std::make_shared<Sphere>(point3D{0, -1000, -1}, 1000, ground));
    world.add(std::make_shared<Sphere>(point3D{0, 1, 2}, 1, glass));
    world.add(std::make_shared<Sphere>(point3D{0, 1, -2}, 1, greyFuzzy));
    world.add(std::make_shared<Sphere>(point3D{2, 1, -6}, 1, greyMetal));
}

// Path Tracer.cpp:199-260. Returns the number of random spheres placed.
inline int addRandomSpheres(HittableList& world, const SceneOptions& opt) {
    std::mt19937 engineA{opt.seedA}, engineB{opt.seedB};
    auto between = [&engineA](double lo, double hi) { return std::uniform_real_distribution<double>{lo, hi}(engineA); };
    int placed = 0;
    const std::size_t fixedCount = world.length();      // objects already there are tested one by one
    PlacementGrid grid(opt.halfExtent);
    auto clips = [&world](const point3D& center, double radius, std::size_t k) {
        return (center - world[k].getCenter()).length() < radius + world[k].minDistanceApart();   // Path Tracer.cpp:222
    };
    for (int i = 0; i < opt.numberOfSpheres; ++i) {
        const double selector = between(0, 1);
        const double radius = between(0.1, 0.4);
        point3D center;
        int attempts = 0;
        bool fits = false;
        while (++attempts <= 50) {
            const double x = between(-opt.halfExtent, opt.halfExtent);
            const double z = between(-opt.halfExtent, opt.halfExtent);
            center = point3D{x, radius, z};
            fits = true;
            const std::size_t scanUpTo = opt.useGrid ? fixedCount : world.length();
            for (std::size_t k = 0; k < scanUpTo; ++k) {
                if (clips(center, radius, k)) { fits = false; break; }
            }
            if (fits && opt.useGrid)
                fits = !grid.anyNear(center, [&](std::size_t k) { return clips(center, radius, k); });
            if (fits) break;
        }
        if (!fits) break;             // no room left: the whole generation ends (Path Tracer.cpp:236)
        std::shared_ptr<Material> material;
        if (selector < opt.diffuseBelow) {
            material = std::make_shared<Lambertian>(randVector(engineB, 0, 1));
        } else if (selector < opt.metalBelow) {
            const colour tint = randVector(engineB, 0.6, 1);
            const double fuzz = between(0, 1) / 2;
            material = std::make_shared<Metal>(tint, fuzz);
        } else {
            material = std::make_shared<Dielectric>(1.5);
        }
        grid.insert(center, world.length());
        world.add(std::make_shared<Sphere>(center, radius, material));
        ++placed;
    }
    return placed;
}

}   // namespace dp
#endif
