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
            for (std::size_t k = 0; k < world.length(); ++k) {
                if ((center - world[k].getCenter()).length() < radius + world[k].minDistanceApart()) { fits = false; break; }
            }
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
        world.add(std::make_shared<Sphere>(center, radius, material));
        ++placed;
    }
    return placed;
}

}   // namespace dp
#endif
