// AABB.h — mirror of Source/AABB.h:12-29 / Source/AABB.cpp.  Construction, union, padding and
// LongestAxis are host utilities; the slab test AABB::Hit (AABB.cpp:38-64) is part of the hot path
// and exists only on the device (prt_device.h, Trav::inner_step) — there is no host version.
#pragma once
#include "Interval.h"
#include "Math.h"
namespace Pooraytracer {
class AABB {
public:
    Interval x, y, z;
    AABB() = default;
    AABB(const Interval& x_, const Interval& y_, const Interval& z_) : x(x_), y(y_), z(z_) { PadToMinimus(); }
    AABB(const vec3& a, const vec3& b) {
        x = (a.x <= b.x) ? Interval(a.x, b.x) : Interval(b.x, a.x);
        y = (a.y <= b.y) ? Interval(a.y, b.y) : Interval(b.y, a.y);
        z = (a.z <= b.z) ? Interval(a.z, b.z) : Interval(b.z, a.z);
        PadToMinimus();
    }
    AABB(const AABB& b0, const AABB& b1) : x(b0.x, b1.x), y(b0.y, b1.y), z(b0.z, b1.z) {}
    const Interval& GetAxisInterval(int axis) const { return axis == 1 ? y : (axis == 2 ? z : x); }
    int LongestAxis() const {
        if (x.Length() > y.Length()) return x.Length() > z.Length() ? 0 : 2;
        return y.Length() > z.Length() ? 1 : 2;
    }
    static const AABB empty, universe;

private:
    void PadToMinimus() {
        const double delta = 0.0001;
        if (x.Length() < delta) x = x.Expand(delta);
        if (y.Length() < delta) y = y.Expand(delta);
        if (z.Length() < delta) z = z.Expand(delta);
    }
};
// correctly initialised regardless of link order (the reference's AABB::empty is not: SURVEY.md §0.3)
inline const AABB AABB::empty = AABB(Interval(), Interval(), Interval());
inline const AABB AABB::universe = AABB(Interval::universe, Interval::universe, Interval::universe);
} // namespace Pooraytracer
