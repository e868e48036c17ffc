// HittableList.h — the list of objects (reference interface: HittableList.h:19-46). isHit is the
// device's linear scan over the flattened list, closest hit, ties to the later index.
#ifndef PT_HOST_HITTABLELIST_H
#define PT_HOST_HITTABLELIST_H

#include <cstdint>
#include <memory>
#include <vector>

#include "Hittable.h"

class HittableList : public Hittable {
    std::vector<std::shared_ptr<Hittable>> m_objectList;
    std::uint64_t m_version = 1;           // bumped by add/clear: the device copy is stale afterwards
    mutable bool m_usedDevice = false;

public:
    HittableList() {}
    HittableList(std::shared_ptr<Hittable> inObject) { add(inObject); }
    ~HittableList() override { if (m_usedDevice) dp::DeviceSession::instance().forgetScene(this); }

    void add(std::shared_ptr<Hittable> inObject) { m_objectList.push_back(inObject); ++m_version; }
    void clear() { m_objectList.clear(); ++m_version; }
    const Hittable& operator[](std::size_t inIndex) const { return *m_objectList[inIndex]; }
    std::size_t length() const { return m_objectList.size(); }
    std::uint64_t version() const { return m_version; }

    bool isHit(const Ray& inRay, double t_min, double t_max, HitRecord& inRec) const override {
        if (m_objectList.empty()) return false;
        std::vector<dp::SphereDesc> flat;
        std::vector<std::shared_ptr<Material>> mats;
        describe(flat, mats);
        dp::DeviceSession::instance().bindScene(this, m_version, flat);
        m_usedDevice = true;
        return dp::deviceIsHit(inRay, t_min, t_max, mats, inRec);
    }

    double minDistanceApart() const override { return 0; }

    dp::PhysicsVector<3> getCenter() const override {
        dp::PhysicsVector<3> center{0, 0, 0};
        for (const auto& h : m_objectList) center += h->getCenter();
        return center.scaledBy(1.0 / static_cast<double>(m_objectList.size()));
    }

    void describe(std::vector<dp::SphereDesc>& out, std::vector<std::shared_ptr<Material>>& materials) const override {
        for (const auto& h : m_objectList) h->describe(out, materials);
    }
};
#endif
