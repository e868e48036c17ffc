// Settings.h — the values main() reads from config.txt, with the reference's defaults and its
// all-or-nothing fallback (Path Tracer.cpp:114-174), plus optional keys of this build whose
// absence reproduces the reference.
#ifndef PT_HOST_SETTINGS_H
#define PT_HOST_SETTINGS_H

#include <cstdint>
#include <iostream>
#include <string>

#include "ConfigReader.h"
#include "VectorFunc.h"

namespace dp {

struct Settings {
    double imageAspectRatio = 16.0 / 9.0;
    int imageWidth = 400;
    PhysicsVector<3> cameraPosition{8, 2, 3};
    PhysicsVector<3> cameraLookingAt{0, 0, 0};
    PhysicsVector<3> cameraUpOrient{0, 1, 0};
    double focalLength = 1;
    double verticalFOV = 60;
    double apertureSize = 0.1;
    // Computed from the DEFAULT camera before the file is read and never updated afterwards
    // (Path Tracer.cpp:127): always sqrt(77).
    double focusDistance = (PhysicsVector<3>{0, 0, 0} - PhysicsVector<3>{8, 2, 3}).length();
    int raysPerPixel = 100;
    int materialMaxDepth = 50;
    bool usedDefaults = false;        // the catch block ran
    // optional keys of this build
    int numberOfSpheres = 250;        // hard-coded in the reference (Path Tracer.cpp:199)
    std::uint32_t sceneSeedA = 1, sceneSeedB = 2;
    std::uint64_t renderSeed = 42;
    double sceneHalfExtent = 10.0;
    int numberOfGPUs = 1;
    bool binaryOutput = false;        // outputFormat = P6 (the reference writes ASCII P3)
    bool gridSearch = false;          // closestHitSearch = grid (the reference tests every sphere: "scan")

    int imageHeight() const { return static_cast<int>(imageWidth / imageAspectRatio); }   // Path Tracer.cpp:176
};

// Reads the ten keys the reference reads, in its order; any ConfigException resets them to the
// defaults exactly as the catch block does (cameraUpOrient is NOT reset there, Path Tracer.cpp:163-172).
inline Settings readSettings(ConfigReader& config, std::ostream* log = nullptr) {
    Settings s;
    try {
        if (log) *log << "Loading settings from configuration file.\n";
        s.imageAspectRatio = config.readValue("imageAspectRatio", s.imageAspectRatio);
        s.imageWidth = config.readValue("imageWidth", s.imageWidth);
        s.raysPerPixel = config.readValue("raysPerPixel", s.raysPerPixel);
        s.materialMaxDepth = config.readValue("materialMaxDepth", s.materialMaxDepth);
        readVector(config.readValue("cameraPosition"), s.cameraPosition);
        readVector(config.readValue("cameraLookingAt"), s.cameraLookingAt);
        readVector(config.readValue("cameraUpOrient"), s.cameraUpOrient);
        s.focalLength = config.readValue("focalLength", s.focalLength);
        s.verticalFOV = config.readValue("verticalFOV", s.verticalFOV);
        s.apertureSize = config.readValue("apertureSize", s.apertureSize);
        if (config.contains("numberOfSpheres")) s.numberOfSpheres = config.readValue("numberOfSpheres", s.numberOfSpheres);
        if (config.contains("sceneSeedA")) s.sceneSeedA = std::uint32_t(config.readValue("sceneSeedA", 1L));
        if (config.contains("sceneSeedB")) s.sceneSeedB = std::uint32_t(config.readValue("sceneSeedB", 2L));
        if (config.contains("renderSeed")) s.renderSeed = std::uint64_t(config.readValue("renderSeed", 42L));
        if (config.contains("sceneHalfExtent")) s.sceneHalfExtent = config.readValue("sceneHalfExtent", s.sceneHalfExtent);
        if (config.contains("numberOfGPUs")) s.numberOfGPUs = config.readValue("numberOfGPUs", s.numberOfGPUs);
        if (config.contains("outputFormat")) s.binaryOutput = config.readValue("outputFormat") == "P6";
        if (config.contains("closestHitSearch")) s.gridSearch = config.readValue("closestHitSearch") == "grid";
        config.clear();
        if (log) *log << "All values read from file correctly.\n";
    } catch (ConfigReader::ConfigException& except) {
        if (log) *log << "Error reading data from config.txt: " << except.what() << '\n' << "Loading default values for those variables.\n";
        const Settings d;
        s.imageAspectRatio = d.imageAspectRatio; s.imageWidth = d.imageWidth;
        s.raysPerPixel = d.raysPerPixel; s.materialMaxDepth = d.materialMaxDepth;
        s.cameraPosition = d.cameraPosition; s.cameraLookingAt = d.cameraLookingAt;
        s.focalLength = d.focalLength; s.verticalFOV = d.verticalFOV; s.apertureSize = d.apertureSize;
        s.usedDefaults = true;
    }
    return s;
}

inline Settings readSettingsFile(const char* path, std::ostream* log = nullptr) {
    ConfigReader config;
    try {
        config.load(path);
    } catch (ConfigReader::ConfigException& except) {
        if (log) *log << "Loading settings from configuration file.\nError reading data from config.txt: " << except.what()
                      << "\nLoading default values for those variables.\n";
        Settings d;
        d.usedDefaults = true;
        return d;
    }
    return readSettings(config, log);
}

}   // namespace dp
#endif
