// DeviceSession.h — the process-wide connection of the host classes to the gfx950 library.
// Every geometric or radiometric answer the host classes give (Hittable::isHit,
// Material::isScattered, Camera::getCurrentRay, the frame itself) is computed by the HIP kernels
// through include/pt_hip.h. There is no host-side arithmetic fallback: without a usable device
// the first call throws.
#ifndef PT_HOST_DEVICESESSION_H
#define PT_HOST_DEVICESESSION_H

#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pt_hip.h"

namespace dp {

struct SphereDesc {        // one flattened Sphere with its Material, in list order
    double cx, cy, cz, r;
    int kind;
    double ar, ag, ab, fuzz, ior;
    const void* material;  // the std::shared_ptr<Material> target, to hand back in a HitRecord
};

class DeviceError : public std::runtime_error {
public:
    explicit DeviceError(const std::string& what) : std::runtime_error(what) {}
};

class DeviceSession {
    pt_handle m_handle = nullptr;
    const void* m_sceneOwner = nullptr;
    std::uint64_t m_sceneVersion = 0;
    const void* m_cameraOwner = nullptr;

    explicit DeviceSession(int device) {
        if (pt_create(device, &m_handle) != 0)
            throw DeviceError(std::string("pt_create: ") + pt_last_error(nullptr));
    }

public:
    DeviceSession(const DeviceSession&) = delete;
    DeviceSession& operator=(const DeviceSession&) = delete;
    ~DeviceSession() { if (m_handle) pt_destroy(m_handle); }

    static DeviceSession& instance(int device = 0) {
        static DeviceSession session(device);
        return session;
    }

    pt_handle handle() const { return m_handle; }

    void check(int rc, const char* what) const {
        if (rc != 0) throw DeviceError(std::string(what) + ": " + pt_last_error(m_handle));
    }

    // Uploads a flattened scene unless (owner, version) is the one already resident.
    void bindScene(const void* owner, std::uint64_t version, const std::vector<SphereDesc>& s) {
        if (owner == m_sceneOwner && version == m_sceneVersion) return;
        uploadScene(m_handle, s);
        m_sceneOwner = owner;
        m_sceneVersion = version;
    }

    void uploadScene(pt_handle h, const std::vector<SphereDesc>& s) const {
        if (s.empty()) throw DeviceError("an empty scene cannot be uploaded");
        const std::size_t n = s.size();
        std::vector<double> cx(n), cy(n), cz(n), r(n), ar(n), ag(n), ab(n), fz(n), ior(n);
        std::vector<int> kind(n);
        for (std::size_t i = 0; i < n; ++i) {
            cx[i] = s[i].cx; cy[i] = s[i].cy; cz[i] = s[i].cz; r[i] = s[i].r; kind[i] = s[i].kind;
            ar[i] = s[i].ar; ag[i] = s[i].ag; ab[i] = s[i].ab; fz[i] = s[i].fuzz; ior[i] = s[i].ior;
        }
        int rc = pt_set_scene(h, int(n), cx.data(), cy.data(), cz.data(), r.data(), kind.data(), ar.data(), ag.data(),
                              ab.data(), fz.data(), ior.data());
        if (rc != 0) throw DeviceError(std::string("pt_set_scene: ") + pt_last_error(h));
    }

    void bindCamera(const void* owner, const double cam19[19]) {
        if (owner == m_cameraOwner) return;
        check(pt_set_camera(m_handle, cam19, cam19 + 3, cam19 + 6, cam19 + 9, cam19 + 12, cam19 + 15, cam19[18]),
              "pt_set_camera");
        m_cameraOwner = owner;
    }
    void forgetCamera(const void* owner) { if (m_cameraOwner == owner) m_cameraOwner = nullptr; }
    void forgetScene(const void* owner) { if (m_sceneOwner == owner) m_sceneOwner = nullptr; }
};

// Uniform numbers for the single-call forms of isScattered / getCurrentRay, which draw their own
// random numbers in the reference too (VectorFunc.cpp:15-19, Dielectric.cpp:52-56). The frame
// renderer does not use this: its numbers are philox blocks drawn on the device.
inline double hostUniform() {
    static thread_local std::mt19937 engine{std::random_device{}()};
    return std::uniform_real_distribution<double>{0.0, 1.0}(engine);
}

}   // namespace dp
#endif
