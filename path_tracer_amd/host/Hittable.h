// Hittable.h — HitRecord and the abstract Hittable (reference interface: Hittable.h:19-51).
// isHit is answered by the device scan (pt_scan); describe() is the export hook the flattener
// needs, because the reference's classes keep their data private.
#ifndef PT_HOST_HITTABLE_H
#define PT_HOST_HITTABLE_H

#include <memory>
#include <vector>

#include "DeviceSession.h"
#include "PhysicsVector.h"
#include "Ray.h"

class Material;

struct HitRecord {
    dp::PhysicsVector<3> m_point;
    dp::PhysicsVector<3> m_normal;              // always opposes the incoming ray
    std::shared_ptr<Material> m_materialPtr;
    double m_interval{0};
    bool m_frontFace{false};

    // Hittable.h:26-31 of the reference: only a sign flip, kept on the host for callers that
    // build records by hand; records coming from isHit are already oriented by the device.
    void setNormalDirection(const Ray& inRay, const dp::PhysicsVector<3>& outwardNormal) {
        m_frontFace = inRay.direction().innerProduct(outwardNormal) < 0;
        m_normal = m_frontFace ? outwardNormal : -outwardNormal;
    }
};

class Hittable {
public:
    virtual ~Hittable() = default;
    virtual bool isHit(const Ray& inRay, double t_min, double t_max, HitRecord& inRec) const = 0;
    virtual double minDistanceApart() const = 0;
    virtual dp::PhysicsVector<3> getCenter() const = 0;
    // Appends this object's spheres, in order, with their materials.
    virtual void describe(std::vector<dp::SphereDesc>& out, std::vector<std::shared_ptr<Material>>& materials) const = 0;
};

namespace dp {
// One ray against a flattened scene that is resident on the device.
inline bool deviceIsHit(const Ray& ray, double t_min, double t_max, const std::vector<std::shared_ptr<Material>>& materials,
                        HitRecord& rec) {
    DeviceSession& s = DeviceSession::instance();
    const double o[3] = {ray.origin().x(), ray.origin().y(), ray.origin().z()};
    const double d[3] = {ray.direction().x(), ray.direction().y(), ray.direction().z()};
    double t = 0, p[3], n[3];
    int idx = -1, front = 0;
    s.check(pt_scan(s.handle(), 1, o, d, t_min, &t, &idx, p, n, &front), "pt_scan");
    if (idx < 0 || t_max < t) return false;
    rec.m_interval = t;
    rec.m_point = PhysicsVector<3>{p[0], p[1], p[2]};
    rec.m_normal = PhysicsVector<3>{n[0], n[1], n[2]};
    rec.m_frontFace = front != 0;
    rec.m_materialPtr = materials[std::size_t(idx)];
    return true;
}
}   // namespace dp
#endif
