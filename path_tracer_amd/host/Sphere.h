// Sphere.h — a sphere with its material (reference interface: Sphere.h:15-46).
#ifndef PT_HOST_SPHERE_H
#define PT_HOST_SPHERE_H

#include <atomic>

#include "Hittable.h"
#include "Material.h"

class Sphere : public Hittable {
    dp::PhysicsVector<3> m_center;
    double m_radius;
    std::shared_ptr<Material> m_material;

public:
    Sphere(dp::PhysicsVector<3> inCenter, double inRadius, std::shared_ptr<Material> inMat)
        : m_center{inCenter}, m_radius{inRadius}, m_material{inMat} {}
    ~Sphere() override { dp::DeviceSession* s = sessionIfAny(); if (s) s->forgetScene(this); }

    bool isHit(const Ray& inRay, double t_min, double t_max, HitRecord& inRec) const override {
        std::vector<dp::SphereDesc> flat;
        std::vector<std::shared_ptr<Material>> mats;
        describe(flat, mats);
        dp::DeviceSession::instance().bindScene(this, 1, flat);
        used() = true;
        return dp::deviceIsHit(inRay, t_min, t_max, mats, inRec);
    }
    double minDistanceApart() const override { return m_radius; }
    dp::PhysicsVector<3> getCenter() const override { return m_center; }

    void describe(std::vector<dp::SphereDesc>& out, std::vector<std::shared_ptr<Material>>& materials) const override {
        const MaterialDesc m = m_material->describe();
        out.push_back(dp::SphereDesc{m_center.x(), m_center.y(), m_center.z(), m_radius, m.kind, m.albedo[0], m.albedo[1],
                                     m.albedo[2], m.fuzz, m.ior, m_material.get()});
        materials.push_back(m_material);
    }

private:
    // The session is only touched by objects that have used the device, so that building and
    // destroying a scene needs no GPU.
    static bool& used() { static bool u = false; return u; }
    static dp::DeviceSession* sessionIfAny() { return used() ? &dp::DeviceSession::instance() : nullptr; }
};
#endif
