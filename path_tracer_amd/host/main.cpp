// pathtracer — the reference's program flow (Path Tracer.cpp:110-324) with the thread pool
// replaced by the MI355X sample loop: read config.txt, build the camera and the scene, render
// on the device(s), write ./outputImage.ppm (ASCII P3, byte-compatible with the reference's).
//
// Usage: pathtracer [config.txt] [outputImage.ppm]
#include <chrono>
#include <fstream>
#include <iostream>
#include <memory>
#include <vector>

#include "Camera.h"
#include "DeviceRenderer.h"
#include "Output.h"
#include "SceneGenerator.h"
#include "Settings.h"

int main(int argc, char** argv) {
    const char* configPath = argc > 1 ? argv[1] : "config.txt";
    const char* outPath = argc > 2 ? argv[2] : "./outputImage.ppm";
    try {
        const dp::Settings s = dp::readSettingsFile(configPath, &std::cout);
        const int width = s.imageWidth, height = s.imageHeight();
        Camera camera(s.cameraPosition, s.cameraLookingAt, s.cameraUpOrient, s.imageAspectRatio, s.focalLength, s.verticalFOV,
                      s.apertureSize, s.focusDistance);

        HittableList world;
        dp::addFixedSpheres(world);
        dp::SceneOptions opt;
        opt.numberOfSpheres = s.numberOfSpheres; opt.seedA = s.sceneSeedA; opt.seedB = s.sceneSeedB; opt.halfExtent = s.sceneHalfExtent;
        const int placed = dp::addRandomSpheres(world, opt);
        std::cout << "Scene: 5 fixed + " << placed << " random spheres.\n";

        // One renderer per GPU, each taking a disjoint range of the samples of every pixel.
        const int gpus = s.numberOfGPUs < 1 ? 1 : s.numberOfGPUs;
        std::vector<std::unique_ptr<dp::DeviceRenderer>> devs;
        for (int g = 0; g < gpus; ++g) {
            devs.push_back(std::make_unique<dp::DeviceRenderer>(g));
            devs.back()->setGridSearch(s.gridSearch);
            devs.back()->setWorld(world);
            devs.back()->setCamera(camera);
            devs.back()->begin(width, height);
        }
        std::vector<pt_handle> hs;
        for (auto& d : devs) hs.push_back(d->handle());
        auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
        double setupSeconds = 0;
        if (gpus > 1) {
            // Communicator set-up (hundreds of milliseconds, once per process) is not part of the frame: the buffers are
            // still all zero here, so summing them changes nothing and leaves the communicators ready.
            const auto ts = std::chrono::steady_clock::now();
            if (pt_reduce(hs.data(), gpus, 0) != 0) throw dp::DeviceError(std::string("pt_reduce: ") + pt_last_error(hs[0]));
            setupSeconds = since(ts);
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (int g = 0; g < gpus; ++g) {
            const int b = int((long long)s.raysPerPixel * g / gpus), e = int((long long)s.raysPerPixel * (g + 1) / gpus);
            devs[g]->accumulate(b, e, s.materialMaxDepth, s.renderSeed);
        }
        for (auto& d : devs) d->synchronize();
        const double renderSeconds = since(t0);
        const auto t1 = std::chrono::steady_clock::now();
        if (gpus > 1 && pt_reduce(hs.data(), gpus, 0) != 0) throw dp::DeviceError(std::string("pt_reduce: ") + pt_last_error(hs[0]));
        const double reduceSeconds = since(t1);
        const auto t2 = std::chrono::steady_clock::now();
        const std::vector<unsigned char> bytes = devs[0]->quantised(s.raysPerPixel);   // output stage on the device
        const double outputSeconds = since(t2);
        const double seconds = since(t0);
        std::uint64_t rays = 0;
        for (auto& d : devs) rays += d->raysTraced();
        const double samples = double(width) * height * s.raysPerPixel;
        std::cout << "Rendered " << width << "x" << height << " at " << s.raysPerPixel << " spp on " << gpus << " GPU(s) in " << seconds
                  << " s: " << samples / seconds / 1e6 << " Msamples/s, " << double(rays) / seconds / 1e9 << " Grays/s"
                  << " (render " << renderSeconds << " s, reduce " << reduceSeconds << " s, gamma+quantise+read-back " << outputSeconds
                  << " s; communicator set-up before the frame " << setupSeconds << " s)\n";

        const auto t3 = std::chrono::steady_clock::now();
        std::ofstream out(outPath, std::ios::out | std::ios::binary);
        if (s.binaryOutput) dp::writeImageP6(out, bytes, width, height);
        else dp::writeImageP3(out, bytes, width, height);                               // byte for byte the reference's file
        out.close();
        std::cout << "Wrote " << outPath << " (" << (s.binaryOutput ? "P6" : "P3") << ") in " << since(t3) << " s\n";
    } catch (const std::exception& e) {
        std::cerr << "pathtracer: " << e.what() << '\n';
        return 1;
    }
    return 0;
}
