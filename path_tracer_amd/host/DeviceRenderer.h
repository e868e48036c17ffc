// DeviceRenderer.h — what the host calls instead of the thread pool and the row tasks
// (Path Tracer.cpp:286-312): one object per GPU that takes the world and the camera, renders a
// sample range of every pixel on the device and hands back the reference's `pixels` vector.
#ifndef PT_HOST_DEVICERENDERER_H
#define PT_HOST_DEVICERENDERER_H

#include <cstdint>
#include <vector>

#include "Camera.h"
#include "DeviceSession.h"
#include "HittableList.h"

namespace dp {

class DeviceRenderer {
    pt_handle m_handle = nullptr;
    int m_width = 0, m_height = 0;
    std::uint64_t m_rays = 0;

    void check(int rc, const char* what) const {
        if (rc != 0) throw DeviceError(std::string(what) + ": " + pt_last_error(m_handle));
    }

public:
    explicit DeviceRenderer(int device = 0) {
        if (pt_create(device, &m_handle) != 0) throw DeviceError(std::string("pt_create: ") + pt_last_error(nullptr));
    }
    DeviceRenderer(const DeviceRenderer&) = delete;
    DeviceRenderer& operator=(const DeviceRenderer&) = delete;
    ~DeviceRenderer() { if (m_handle) pt_destroy(m_handle); }

    pt_handle handle() const { return m_handle; }

    void setWorld(const HittableList& world) {
        std::vector<SphereDesc> flat;
        std::vector<std::shared_ptr<Material>> mats;
        world.describe(flat, mats);
        DeviceSession::instance().uploadScene(m_handle, flat);
    }
    // Closest-hit search: false = every sphere for every ray (HittableList.cpp:27, the default), true = the
    // uniform grid of pt_set_accel (same answer, far fewer tests in scenes of thousands of spheres).
    void setGridSearch(bool on) { check(pt_set_accel(m_handle, on ? PT_ACCEL_GRID : PT_ACCEL_NONE), "pt_set_accel"); }
    void setCamera(const Camera& camera) {
        double c[19];
        camera.deviceConstants(c);
        check(pt_set_camera(m_handle, c, c + 3, c + 6, c + 9, c + 12, c + 15, c[18]), "pt_set_camera");
    }

    void begin(int width, int height) {
        check(pt_clear_accum(m_handle, width, height), "pt_clear_accum");
        m_width = width; m_height = height;
    }
    // Adds samples [sppBegin, sppEnd) of every pixel; asynchronous.
    void accumulate(int sppBegin, int sppEnd, int maxDepth, std::uint64_t seed) {
        check(pt_render(m_handle, m_width, m_height, sppBegin, sppEnd, maxDepth, seed, nullptr, nullptr), "pt_render");
    }
    // The sums, top row first, as the reference's `pixels` (Path Tracer.cpp:292-298).
    std::vector<PhysicsVector<3>> pixels() {
        std::vector<float> f(std::size_t(m_width) * m_height * 3);
        check(pt_read_accum(m_handle, m_width, m_height, f.data()), "pt_read_accum");
        std::vector<PhysicsVector<3>> out(std::size_t(m_width) * m_height);
        for (std::size_t i = 0; i < out.size(); ++i) out[i] = PhysicsVector<3>{f[3 * i], f[3 * i + 1], f[3 * i + 2]};
        check(pt_ray_count(m_handle, &m_rays), "pt_ray_count");
        return out;
    }
    // Gamma, clamp and quantise on the device (writeColour, Path Tracer.cpp:62-74): r g b bytes, top row first.
    std::vector<unsigned char> quantised(int samplesPerPixel) {
        std::vector<unsigned char> out(std::size_t(m_width) * m_height * 3);
        check(pt_quantise_rgb8(m_handle, m_width, m_height, samplesPerPixel, nullptr, out.data()), "pt_quantise_rgb8");
        check(pt_ray_count(m_handle, &m_rays), "pt_ray_count");
        return out;
    }
    // Resume a frame from saved sums (the float buffer read with pixels() / pt_read_accum).
    void resume(int width, int height, const std::vector<float>& sums) {
        if (sums.size() != std::size_t(width) * height * 3) throw DeviceError("resume: buffer size does not match the image");
        check(pt_write_accum(m_handle, width, height, sums.data()), "pt_write_accum");
        m_width = width; m_height = height;
    }
    // Rays traced since the renderer was created (synchronises with the device).
    std::uint64_t raysTraced() {
        check(pt_ray_count(m_handle, &m_rays), "pt_ray_count");
        return m_rays;
    }
    void synchronize() { check(pt_synchronize(m_handle), "pt_synchronize"); }

    // Milliseconds the sample-loop kernel ran for, summed over the renders since the last call.
    double kernelMilliseconds() {
        std::vector<float> ms(8192);
        int n = 0;
        check(pt_kernel_times(m_handle, ms.data(), int(ms.size()), &n), "pt_kernel_times");
        double s = 0;
        for (int i = 0; i < n && i < int(ms.size()); ++i) s += ms[i];
        return s;
    }
};

}   // namespace dp
#endif
