// Interval.h — mirror of the reference's Interval (Source/Interval.h:8-39): same members and semantics.
#pragma once
#include <limits>
namespace Pooraytracer {
class Interval {
public:
    double min, max;
    Interval() : min(+std::numeric_limits<double>::infinity()), max(-std::numeric_limits<double>::infinity()) {}
    Interval(double min_, double max_) : min(min_), max(max_) {}
    Interval(const Interval& a, const Interval& b) {
        min = a.min <= b.min ? a.min : b.min;
        max = a.max >= b.max ? a.max : b.max;
    }
    double Length() const { return max - min; }
    bool Contains(double x) const { return min <= x && x <= max; }
    bool Surrounds(double x) const { return min < x && x < max; }
    double Clamp(double x) const { return x < min ? min : (x > max ? max : x); }
    Interval Expand(double delta) const {
        double padding = delta / 2.;
        return Interval(min - padding, max + padding);
    }
    static const Interval empty, universe;
};
inline const Interval Interval::empty = Interval(+std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity());
inline const Interval Interval::universe = Interval(-std::numeric_limits<double>::infinity(), +std::numeric_limits<double>::infinity());
inline Interval operator+(const Interval& i, double d) { return Interval(i.min + d, i.max + d); }
inline Interval operator+(double d, const Interval& i) { return i + d; }
} // namespace Pooraytracer
