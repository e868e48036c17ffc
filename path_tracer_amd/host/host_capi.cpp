// host_capi.cpp — C entry points over the host-side C++ (scene generator, camera set-up, config
// reader, PPM writer, and the device-backed class surface) so that the Python tests can drive them.
#include <cstring>
#include <sstream>
#include <string>

#include "Camera.h"
#include "DeviceRenderer.h"
#include "Output.h"
#include "SceneGenerator.h"
#include "Settings.h"

namespace {
thread_local std::string g_err;
int flatten_out(const HittableList& world, int cap, double* cx, double* cy, double* cz, double* r, int* kind, double* ar,
                double* ag, double* ab, double* fuzz, double* ior) {
    std::vector<dp::SphereDesc> flat;
    std::vector<std::shared_ptr<Material>> mats;
    world.describe(flat, mats);
    const int n = int(flat.size());
    for (int i = 0; i < n && i < cap; ++i) {
        cx[i] = flat[i].cx; cy[i] = flat[i].cy; cz[i] = flat[i].cz; r[i] = flat[i].r; kind[i] = flat[i].kind;
        ar[i] = flat[i].ar; ag[i] = flat[i].ag; ab[i] = flat[i].ab; fuzz[i] = flat[i].fuzz; ior[i] = flat[i].ior;
    }
    return n;
}
HittableList build_world(int n, const double* cx, const double* cy, const double* cz, const double* r, const int* kind,
                         const double* ar, const double* ag, const double* ab, const double* fuzz, const double* ior) {
    HittableList world;
    for (int i = 0; i < n; ++i) {
        std::shared_ptr<Material> m;
        if (kind[i] == PT_LAMBERTIAN) m = std::make_shared<Lambertian>(dp::colour{ar[i], ag[i], ab[i]});
        else if (kind[i] == PT_METAL) m = std::make_shared<Metal>(dp::colour{ar[i], ag[i], ab[i]}, fuzz[i]);
        else m = std::make_shared<Dielectric>(ior[i]);
        world.add(std::make_shared<Sphere>(dp::point3D{cx[i], cy[i], cz[i]}, r[i], m));
    }
    return world;
}
}   // namespace

extern "C" {

const char* pth_last_error() { return g_err.c_str(); }

// Five fixed + up to `count` random spheres, flattened in list order. Returns the sphere count.
int pth_generate_scene(int count, unsigned seedA, unsigned seedB, double half_extent, double diffuse_below,
                       double metal_below, int use_grid, int cap, double* cx, double* cy, double* cz, double* r, int* kind, double* ar,
                       double* ag, double* ab, double* fuzz, double* ior) {
    HittableList world;
    dp::addFixedSpheres(world);
    dp::SceneOptions opt;
    opt.numberOfSpheres = count; opt.seedA = seedA; opt.seedB = seedB; opt.halfExtent = half_extent;
    opt.diffuseBelow = diffuse_below; opt.metalBelow = metal_below; opt.useGrid = use_grid != 0;
    dp::addRandomSpheres(world, opt);
    return flatten_out(world, cap, cx, cy, cz, r, kind, ar, ag, ab, fuzz, ior);
}

void pth_camera_setup(const double* pos, const double* look, const double* up, double aspect, double focal, double fov,
                      double aperture, double focus, double* cam19) {
    Camera c(dp::point3D{pos[0], pos[1], pos[2]}, dp::point3D{look[0], look[1], look[2]}, dp::point3D{up[0], up[1], up[2]},
             aspect, focal, fov, aperture, focus);
    c.deviceConstants(cam19);
}

// Parses config text as main() does. out[0..16] = aspect, width, spp, depth, pos(3), look(3), up(3), focal, fov,
// aperture, focus distance; out[17] = height; out[18..25] = the optional keys of this build: numberOfSpheres,
// sceneSeedA, sceneSeedB, renderSeed, sceneHalfExtent, numberOfGPUs, outputFormat == P6, closestHitSearch == grid.
// Returns 1 when the all-defaults fallback ran, else 0.
int pth_read_settings(const char* text, double* out) {
    dp::Settings s;
    try {
        dp::ConfigReader cfg;
        cfg.parse(text);
        s = dp::readSettings(cfg, nullptr);
    } catch (dp::ConfigReader::ConfigException&) {
        s = dp::Settings{};
        s.usedDefaults = true;
    }
    out[0] = s.imageAspectRatio; out[1] = s.imageWidth; out[2] = s.raysPerPixel; out[3] = s.materialMaxDepth;
    for (int i = 0; i < 3; ++i) { out[4 + i] = s.cameraPosition[i]; out[7 + i] = s.cameraLookingAt[i]; out[10 + i] = s.cameraUpOrient[i]; }
    out[13] = s.focalLength; out[14] = s.verticalFOV; out[15] = s.apertureSize; out[16] = s.focusDistance; out[17] = s.imageHeight();
    out[18] = s.numberOfSpheres; out[19] = s.sceneSeedA; out[20] = s.sceneSeedB; out[21] = double(s.renderSeed);
    out[22] = s.sceneHalfExtent; out[23] = s.numberOfGPUs; out[24] = s.binaryOutput ? 1 : 0; out[25] = s.gridSearch ? 1 : 0;
    return s.usedDefaults ? 1 : 0;
}

int pth_read_vector(const char* text, double* out3) {
    dp::PhysicsVector<3> v{9, 9, 9};
    const bool ok = dp::readVector(std::string_view(text), v);
    out3[0] = v.x(); out3[1] = v.y(); out3[2] = v.z();
    return ok ? 1 : 0;
}

// The P3 file for sums over spp samples (top row first). Returns bytes written, or -needed.
long pth_format_ppm(const float* sum, int width, int height, int spp, char* out, long cap) {
    std::vector<dp::PhysicsVector<3>> px(std::size_t(width) * height);
    for (std::size_t i = 0; i < px.size(); ++i) px[i] = dp::PhysicsVector<3>{sum[3 * i], sum[3 * i + 1], sum[3 * i + 2]};
    std::ostringstream os;
    dp::writeImage(os, px, width, height, spp);
    const std::string s = os.str();
    if (long(s.size()) > cap) return -long(s.size());
    std::memcpy(out, s.data(), s.size());
    return long(s.size());
}

// ---- the device-backed class surface (needs a GPU) ----

// HittableList::isHit for one ray. Returns 1 hit / 0 miss / -1 error.
int pth_world_is_hit(int n, const double* cx, const double* cy, const double* cz, const double* r, const int* kind,
                     const double* ar, const double* ag, const double* ab, const double* fuzz, const double* ior,
                     const double* o, const double* d, double t_min, double t_max, double* t, double* p, double* nrm,
                     int* front, int* mat_kind) {
    try {
        HittableList world = build_world(n, cx, cy, cz, r, kind, ar, ag, ab, fuzz, ior);
        HitRecord rec;
        const bool hit = world.isHit(Ray(dp::point3D{o[0], o[1], o[2]}, dp::point3D{d[0], d[1], d[2]}), t_min, t_max, rec);
        if (!hit) return 0;
        *t = rec.m_interval; *front = rec.m_frontFace ? 1 : 0;
        for (int i = 0; i < 3; ++i) { p[i] = rec.m_point[i]; nrm[i] = rec.m_normal[i]; }
        *mat_kind = rec.m_materialPtr->describe().kind;
        return 1;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// Sphere::isHit for one sphere and one ray.
int pth_sphere_is_hit(const double* c, double r, const double* o, const double* d, double t_min, double t_max, double* t,
                      double* p, double* nrm, int* front) {
    try {
        Sphere s(dp::point3D{c[0], c[1], c[2]}, r, std::make_shared<Lambertian>(dp::colour{0.5, 0.5, 0.5}));
        HitRecord rec;
        if (!s.isHit(Ray(dp::point3D{o[0], o[1], o[2]}, dp::point3D{d[0], d[1], d[2]}), t_min, t_max, rec)) return 0;
        *t = rec.m_interval; *front = rec.m_frontFace ? 1 : 0;
        for (int i = 0; i < 3; ++i) { p[i] = rec.m_point[i]; nrm[i] = rec.m_normal[i]; }
        return 1;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// Material::isScattered with given uniforms. Returns the bool, or -1 on error.
int pth_material_scatter(int kind, const double* albedo, double fuzz, double ior, const double* d_in, const double* p,
                         const double* nrm, int front, const double* u, double* d_out, double* atten) {
    try {
        std::shared_ptr<Material> m;
        if (kind == PT_LAMBERTIAN) m = std::make_shared<Lambertian>(dp::colour{albedo[0], albedo[1], albedo[2]});
        else if (kind == PT_METAL) m = std::make_shared<Metal>(dp::colour{albedo[0], albedo[1], albedo[2]}, fuzz);
        else m = std::make_shared<Dielectric>(ior);
        HitRecord rec;
        rec.m_point = dp::point3D{p[0], p[1], p[2]}; rec.m_normal = dp::point3D{nrm[0], nrm[1], nrm[2]};
        rec.m_frontFace = front != 0; rec.m_materialPtr = m;
        dp::colour a; Ray out;
        const bool ok = m->isScatteredWith(Ray(rec.m_point, dp::point3D{d_in[0], d_in[1], d_in[2]}), rec, u, a, out);
        for (int i = 0; i < 3; ++i) { d_out[i] = out.direction()[i]; atten[i] = a[i]; }
        return ok ? 1 : 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// Camera::getCurrentRay with the lens point given.
int pth_camera_ray(const double* pos, const double* look, const double* up, double aspect, double focal, double fov,
                   double aperture, double focus, double x, double y, double diskx, double disky, double* o, double* d) {
    try {
        Camera c(dp::point3D{pos[0], pos[1], pos[2]}, dp::point3D{look[0], look[1], look[2]}, dp::point3D{up[0], up[1], up[2]},
                 aspect, focal, fov, aperture, focus);
        const Ray r = c.getCurrentRay(x, y, diskx, disky);
        for (int i = 0; i < 3; ++i) { o[i] = r.origin()[i]; d[i] = r.direction()[i]; }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// The whole frame through the host classes: settings text -> scene -> device -> P3 bytes.
long pth_render_ppm(const char* config_text, char* out, long cap, double* msamples_per_s) {
    try {
        dp::ConfigReader cfg;
        dp::Settings s;
        try { cfg.parse(config_text); s = dp::readSettings(cfg, nullptr); }
        catch (dp::ConfigReader::ConfigException&) { s = dp::Settings{}; s.usedDefaults = true; }
        Camera camera(s.cameraPosition, s.cameraLookingAt, s.cameraUpOrient, s.imageAspectRatio, s.focalLength, s.verticalFOV,
                      s.apertureSize, s.focusDistance);
        HittableList world;
        dp::addFixedSpheres(world);
        dp::SceneOptions opt;
        opt.numberOfSpheres = s.numberOfSpheres; opt.seedA = s.sceneSeedA; opt.seedB = s.sceneSeedB; opt.halfExtent = s.sceneHalfExtent;
        dp::addRandomSpheres(world, opt);
        dp::DeviceRenderer dev(0);
        dev.setGridSearch(s.gridSearch);
        dev.setWorld(world); dev.setCamera(camera);
        const int w = s.imageWidth, h = s.imageHeight();
        dev.begin(w, h);
        dev.accumulate(0, s.raysPerPixel, s.materialMaxDepth, s.renderSeed);
        auto px = dev.pixels();
        const double ms = dev.kernelMilliseconds();
        if (msamples_per_s) *msamples_per_s = ms > 0 ? double(w) * h * s.raysPerPixel / (ms * 1e-3) / 1e6 : 0;
        std::ostringstream os;
        dp::writeImage(os, px, w, h, s.raysPerPixel);
        const std::string str = os.str();
        if (long(str.size()) > cap) return -long(str.size());
        std::memcpy(out, str.data(), str.size());
        return long(str.size());
    } catch (const std::exception& e) { g_err = e.what(); return 0; }
}

}   // extern "C"
