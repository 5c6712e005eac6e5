// pt_oracle.cpp — CPU oracle for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// PARITY UNPINNED (no reference tests/fixtures exist; reference unbuildable here — see oracle.h).
//
// A plain fp64 restatement of the reference algorithm, function by function.  Every function cites
// the reference file:line it follows (paths relative to the reference repo root).  The arithmetic
// order of every expression follows the reference source (and glm's definitions of dot / cross /
// normalize / length) so that rounding agrees with a g++ -O2 x86-64 build of the reference.
//
// Deliberate, documented departures:
//   B1  AABB::empty is correctly initialised ([+inf,-inf]) — the "Interval.cpp linked first" order.
//   B2  std::rand() is replaced by a keyed counter RNG (seed, pixel, sample) -> 31-bit values, so
//       xi = r / 2^31 has the same granularity as glibc rand()/(RAND_MAX+1.0).
//   B9  shadow ray that escapes the scene: the reference reads an uninitialised HitRecord
//       (Camera.cpp:150-155); defined here as "unoccluded".
//   B13 Phong Scatter leaves `attenuation` unassigned on a bad sample (Material.h:280-282); defined 0.
//   B18 all rows are rendered and the framebuffer is cleared per frame.
//   B20 glm::dvec2(RandomDouble(), RandomDouble()): g++ evaluates right-to-left, so u.y = first
//       draw, u.x = second draw (RandomNumberGenerator.h:59, Material.h:443).
//
// Build: g++ -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).

#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------ glm::dvec3 subset
struct V2 {
    double x, y;
};
struct V3 {
    double x, y, z;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator*(double s, V2 a) { return {s * a.x, s * a.y}; }
// glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// glm::cross
inline V3 cross(V3 x, V3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline double length(V3 v) { return std::sqrt(dot(v, v)); }
// glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
inline V3 normalize(V3 v) { return v * (1.0 / std::sqrt(dot(v, v))); }
inline bool anynan(V3 v) { return v.x != v.x || v.y != v.y || v.z != v.z; }

const double kInf = std::numeric_limits<double>::infinity();
// RandomNumberGenerator.h:10-14
const double Pi = 3.14159265358979323846;
const double InvPi = 0.31830988618379067154;
const double Inv2Pi = 0.15915494309189533577;
const double PiOver2 = 1.57079632679489661923;
const double PiOver4 = 0.78539816339744830961;

// ------------------------------------------------------------------ keyed RNG (departure B2)
inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
struct Rng {
    uint64_t s;
    void seed(uint64_t seed, uint64_t pixel, uint64_t sample) {
        // the seed is hashed on its own, then combined with (pixel, sample) and hashed again (Stafford's mix13 avalanches
        // every input bit; pixel and sample are below 2^32): streams of different seeds are unrelated
        s = mix64(mix64(seed + 0x9E3779B97F4A7C15ULL) ^ ((pixel + 1) << 32) ^ (sample + 1));
        if (s == 0) s = 0x9E3779B97F4A7C15ULL; // the all-zero state is the generator's fixed point
    }
    // RandomDouble(), RandomNumberGenerator.h:16-19: rand() / (RAND_MAX + 1.0), RAND_MAX = 2^31-1.
    // The stream of one (seed, pixel, sample) key is xoroshiro64* (Blackman & Vigna) started from the hashed key:
    // one 32-bit multiply per number; its top 31 bits are used (the low bits are the generator's weak ones).
    double next() {
        uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
        const uint32_t r = s0 * 0x9E3779BBu;
        s1 ^= s0;
        s0 = ((s0 << 26) | (s0 >> 6)) ^ s1 ^ (s1 << 9);
        s1 = (s1 << 13) | (s1 >> 19);
        s = ((uint64_t)s1 << 32) | s0;
        return (double)(r >> 1) / 2147483648.0;
    }
};
thread_local Rng g_rng;
inline double RandomDouble() { return g_rng.next(); }

// ------------------------------------------------------------------ Interval (Interval.h:8-39)
struct Interval {
    double min, max;
    Interval() : min(+kInf), max(-kInf) {}
    Interval(double mn, double mx) : min(mn), max(mx) {}
    Interval(const Interval& a, const Interval& b) { // Interval.h:14-17
        min = a.min <= b.min ? a.min : b.min;
        max = a.max >= b.max ? a.max : b.max;
    }
    double Length() const { return max - min; }
    bool Contains(double x) const { return min <= x && x <= max; } // inclusive, Interval.h:21-23
    Interval Expand(double delta) const {                         // Interval.h:33-36
        double padding = delta / 2.;
        return Interval(min - padding, max + padding);
    }
};

// ------------------------------------------------------------------ Ray (Ray.h:7-17)
struct Ray {
    V3 origin, direction;
    V3 at(double t) const { return origin + direction * t; }
};

// ------------------------------------------------------------------ AABB (AABB.h, AABB.cpp)
struct AABB {
    Interval x, y, z;
    AABB() {}
    AABB(const Interval& x_, const Interval& y_, const Interval& z_) : x(x_), y(y_), z(z_) { // AABB.cpp:10-14
        PadToMinimus();
    }
    AABB(V3 a, V3 b) { // AABB.cpp:16-22
        x = (a.x <= b.x) ? Interval(a.x, b.x) : Interval(b.x, a.x);
        y = (a.y <= b.y) ? Interval(a.y, b.y) : Interval(b.y, a.y);
        z = (a.z <= b.z) ? Interval(a.z, b.z) : Interval(b.z, a.z);
        PadToMinimus();
    }
    AABB(const AABB& b0, const AABB& b1) { // AABB.cpp:24-29 (no padding)
        x = Interval(b0.x, b1.x);
        y = Interval(b0.y, b1.y);
        z = Interval(b0.z, b1.z);
    }
    const Interval& Axis(int a) const { return a == 1 ? y : (a == 2 ? z : x); } // AABB.cpp:31-36
    // AABB.cpp:38-64 — slab test, 1.0/direction[i] recomputed per call, strict `t.max <= t.min` reject
    bool Hit(const Ray& ray, Interval t) const {
        for (int i = 0; i < 3; ++i) {
            const Interval& axis = Axis(i);
            const double inv = 1.0 / ray.direction[i];
            double t0 = (axis.min - ray.origin[i]) * inv;
            double t1 = (axis.max - ray.origin[i]) * inv;
            if (t0 < t1) {
                if (t0 > t.min) t.min = t0;
                if (t1 < t.max) t.max = t1;
            } else {
                if (t1 > t.min) t.min = t1;
                if (t0 < t.max) t.max = t0;
            }
            if (t.max <= t.min) return false;
        }
        return true;
    }
    int LongestAxis() const { // AABB.cpp:66-74
        if (x.Length() > y.Length()) return x.Length() > z.Length() ? 0 : 2;
        return y.Length() > z.Length() ? 1 : 2;
    }
    void PadToMinimus() { // AABB.cpp:76-82
        double delta = 0.0001;
        if (x.Length() < delta) x = x.Expand(delta);
        if (y.Length() < delta) y = y.Expand(delta);
        if (z.Length() < delta) z = z.Expand(delta);
    }
};
// AABB::empty, correctly initialised (departure B1): AABB(Interval::empty x3) -> padding keeps +-inf.
AABB EmptyBox() { return AABB(Interval(), Interval(), Interval()); }

// ------------------------------------------------------------------ HitRecord (Hittable.h:17-28)
struct HitRecord {
    V3 position{0, 0, 0};
    double time = 0;
    V3 normal{0, 0, 0};
    V3 tangent{0, 0, 0};
    V2 uv{0, 0};
    int material = -1;
    bool bFrontFace = false;
    // extras for test export (not in the reference)
    int prim = -1;
    double alpha = 0, beta = 0;
    void SetFaceNormal(const Ray& ray, V3 outward) { // Hittable.cpp:8-13
        bFrontFace = dot(ray.direction, outward) < 0.;
        normal = bFrontFace ? outward : -outward;
    }
};

// ------------------------------------------------------------------ Texture (Texture.h, Texture.cpp:22-71)
struct Texture {
    int width = 0, height = 0, channels = 0;
    std::vector<unsigned char> data;
    static double SRGBToLinear(double c) { // Texture.cpp:66-70
        if (c <= 0.04045) return c * (1. / 12.92);
        return std::pow((c + 0.055) * (1. / 1.055), 2.4);
    }
    V3 GetPixel(int x, int y) const { // Texture.cpp:50-65
        const double colorScale = 1.0 / 255.0;
        int idx = (y * width + x) * channels;
        if (channels >= 3) {
            return {SRGBToLinear(colorScale * data[idx]), SRGBToLinear(colorScale * data[idx + 1]),
                    SRGBToLinear(colorScale * data[idx + 2])};
        }
        double g = colorScale * data[idx];
        return {g, g, g};
    }
    V3 Value(double u, double v) const { // Texture.cpp:22-49
        if (data.empty()) return {0., 1., 1.};
        u = std::clamp(u, 0.0, 1.0);
        v = std::clamp(v, 0.0, 1.0);
        double x = u * (width - 1.);
        double y = (1. - v) * (height - 1.);
        int x0 = static_cast<int>(x);
        int y0 = static_cast<int>(y);
        int x1 = std::min(x0 + 1, width - 1);
        int y1 = std::min(y0 + 1, height - 1);
        double tx = x - x0;
        double ty = y - y0;
        V3 c00 = GetPixel(x0, y0);
        V3 c10 = GetPixel(x1, y0);
        V3 c01 = GetPixel(x0, y1);
        V3 c11 = GetPixel(x1, y1);
        V3 c0 = c00 * (1 - tx) + c10 * tx;
        V3 c1 = c01 * (1 - tx) + c11 * tx;
        return c0 * (1 - ty) + c1 * ty;
    }
};

// ------------------------------------------------------------------ MaterialUtils.h
struct Complex { // MaterialUtils.h:6-44
    double re, im;
    Complex(double r) : re(r), im(0) {}
    Complex(double r, double i) : re(r), im(i) {}
    Complex operator+(Complex z) const { return {re + z.re, im + z.im}; }
    Complex operator-(Complex z) const { return {re - z.re, im - z.im}; }
    Complex operator*(Complex z) const { return {re * z.re - im * z.im, re * z.im + im * z.re}; }
    Complex operator/(Complex z) const {
        double scale = 1 / (z.re * z.re + z.im * z.im);
        return {scale * (re * z.re + im * z.im), scale * (im * z.re - re * z.im)};
    }
};
inline Complex operator+(double v, Complex z) { return Complex(v) + z; }
inline Complex operator-(double v, Complex z) { return Complex(v) - z; }
inline Complex operator*(double v, Complex z) { return Complex(v) * z; }
inline Complex operator/(double v, Complex z) { return Complex(v) / z; }
inline double Norm(const Complex& z) { return z.re * z.re + z.im * z.im; } // :47-49
inline Complex Sqrt(const Complex& z) {                                     // :54-65
    double n = std::sqrt(Norm(z)), t1 = std::sqrt(.5 * (n + std::fabs(z.re))), t2 = .5 * z.im / t1;
    if (n == 0) return 0;
    if (z.re >= 0) return {t1, t2};
    return {std::fabs(t2), std::copysign(t1, z.im)};
}
inline double Clamp(double val, double low, double high) { // :67-71
    if (val < low) return low;
    if (val > high) return high;
    return val;
}
inline double Sqr(double v) { return v * v; }
inline double Cos2Theta(V3 w) { return Sqr(w.z); }
inline double Sin2Theta(V3 w) { return std::max<double>(0., 1 - Cos2Theta(w)); }
inline double Tan2Theta(V3 w) { return Sin2Theta(w) / Cos2Theta(w); }
inline double SinTheta(V3 w) { return std::sqrt(Sin2Theta(w)); }
inline double AbsCosTheta(V3 w) { return std::abs(w.z); }
inline double CosPhi(V3 w) { // :84-87
    double sinTheta = SinTheta(w);
    return (sinTheta == 0) ? 1 : Clamp(w.x / sinTheta, -1, 1);
}
inline double SinPhi(V3 w) { // :88-91
    double sinTheta = SinTheta(w);
    return (sinTheta == 0) ? 0 : Clamp(w.y / sinTheta, -1, 1);
}
inline double AbsDot(V3 a, V3 b) { return std::fabs(dot(a, b)); }
inline double Lerp(double x, double a, double b) { return (1 - x) * a + x * b; }
inline double LengthSquared(V3 v) { return Sqr(v.x) + Sqr(v.y) + Sqr(v.z); }
inline double FrComplex(double cosTheta_i, Complex eta) { // MaterialUtils.h:100-111
    cosTheta_i = Clamp(cosTheta_i, 0, 1);
    double sin2Theta_i = 1 - Sqr(cosTheta_i);
    Complex sin2Theta_t = sin2Theta_i / (eta * eta);
    Complex cosTheta_t = Sqrt(1 - sin2Theta_t);
    Complex r_parl = (eta * cosTheta_i - cosTheta_t) / (eta * cosTheta_i + cosTheta_t);
    Complex r_perp = (cosTheta_i - eta * cosTheta_t) / (cosTheta_i + eta * cosTheta_t);
    return (Norm(r_parl) + Norm(r_perp)) / 2;
}
inline bool SameHemisphere(V3 w, V3 wp) { return w.z * wp.z > 0; }

// ------------------------------------------------------------------ samplers (RandomNumberGenerator.h)
inline V2 SampleUniformDiskConcentric(V2 u) { // :39-56
    V2 uOffset = 2. * u - V2{1., 1.};
    if (uOffset.x == 0. && uOffset.y == 0.) return {0., 0.};
    double theta, r;
    if (std::fabs(uOffset.x) > std::fabs(uOffset.y)) {
        r = uOffset.x;
        theta = PiOver4 * (uOffset.y / uOffset.x);
    } else {
        r = uOffset.y;
        theta = PiOver2 - PiOver4 * (uOffset.x / uOffset.y);
    }
    return r * V2{std::cos(theta), std::sin(theta)};
}
inline V3 SampleCosineHemisphere() { // :57-64, argument order per departure B20
    double first = RandomDouble();
    double second = RandomDouble();
    V2 u{second, first};
    V2 d = SampleUniformDiskConcentric(u);
    double z = std::sqrt(std::max(0.0, 1. - d.x * d.x - d.y * d.y));
    return {d.x, d.y, z};
}
inline V2 SampleUniformDiskPolar(V2 u) { // :69-73
    double r = std::sqrt(u.x);
    double theta = 2 * Pi * u.y;
    return {r * std::cos(theta), r * std::sin(theta)};
}

// ------------------------------------------------------------------ Material (Material.h)
enum { MAT_LAMBERTIAN = 0, MAT_PHONG, MAT_MIRROR, MAT_COOKTORRANCE, MAT_DIFFUSE_LIGHT, MAT_DEBUG, MAT_EMPTY };

struct MaterialEvalContext { // Material.h:21-29
    V3 p;
    V2 uv;
    V3 wo, n, dpdus;
};
struct MaterialSampleContext { // Material.h:39-46; `flags` Unset = 0
    V3 wi{0, 0, 0}, wm{0, 0, 0}, f{0, 0, 0};
    double pdf = 0;
    unsigned flags = 0;
};

struct Material {
    int type = MAT_LAMBERTIAN;
    const Texture* tex = nullptr; // Kd map (Phong: also Ks, Material.h:178-181)
    V3 Kd{0, 0, 0}, Ks{0, 0, 0};
    double Ns = 0, pkd = 1, pks = 0;
    V3 emission{0, 0, 0};
    V3 eta{1, 1, 1}, k{0, 0, 0};
    double alphaX = 0.3, alphaY = 0.3;

    void SetProbabilitiesByNs() { // Material.h:318-327
        if (Ns <= 9.) { pkd = 1.0; pks = 0.0; } else { pkd = 0.6; pks = 0.4; }
    }
    bool HasEmission() const { return type == MAT_DIFFUSE_LIGHT || type == MAT_DEBUG; } // :168,:527
    V3 GetEmission() const { return type == MAT_DIFFUSE_LIGHT ? emission : (type == MAT_DEBUG ? Kd : V3{0, 0, 0}); }
    bool SkipLightSampling() const { // Material.h:73,328,365,539
        switch (type) {
        case MAT_PHONG: return Ns > 1.;
        case MAT_MIRROR: return true;
        case MAT_EMPTY: return true;
        default: return false;
        }
    }
    V3 KdValue(const MaterialEvalContext& c) const { return tex ? tex->Value(c.uv.x, c.uv.y) : Kd; }
    V3 KsValue(const MaterialEvalContext& c) const { return tex ? tex->Value(c.uv.x, c.uv.y) : Ks; }

    // Material.h:76-98
    static V3 LocalToWorld(V3 local, const MaterialEvalContext& c) {
        V3 bitangent = cross(c.dpdus, c.n);
        return normalize(local.x * c.dpdus + local.y * bitangent + local.z * c.n);
    }
    static V3 WorldToLocal(V3 world, const HitRecord& r) {
        V3 bitangent = cross(r.tangent, r.normal);
        double x = dot(world, r.tangent);
        double y = dot(world, bitangent);
        double z = dot(world, r.normal);
        return {x, y, z};
    }
    static V3 Reflect(V3 wo, V3 n) { return -wo + 2. * dot(wo, n) * n; }

    // ---- PhoneReflectance helpers (Material.h:249-262, 299-311)
    double SpecularPDF(V3 wi, const MaterialEvalContext& c) const {
        if (wi.z <= 0.) return 0.0;
        V3 localReflect = normalize(Reflect(c.wo, V3{0., 0., 1.}));
        double cosAlpha = dot(wi, localReflect);
        return (Ns + 1.0) * Inv2Pi * std::pow(cosAlpha, Ns);
    }
    V3 ReflectiveSpaceToLocal(V3 reflect, const MaterialEvalContext& c) const {
        V3 localR = normalize(Reflect(c.wo, V3{0., 0., 1.}));
        V3 V = (std::fabs(localR.x) > 0.9 ? V3{0., 1., 0.} : V3{1., 0., 0.});
        V3 T = normalize(cross(V, localR));
        V3 B = cross(localR, T);
        return reflect.x * T + reflect.y * B + reflect.z * localR;
    }

    // ---- CookTorrance helpers (Material.h:373-435)
    double D(V3 wm) const {
        double tan2Theta = Tan2Theta(wm);
        if (std::isinf(tan2Theta)) return 0;
        double cos4Theta = Sqr(Cos2Theta(wm));
        double e = tan2Theta * (Sqr(CosPhi(wm) / alphaX) + Sqr(SinPhi(wm) / alphaY));
        return 1 / (Pi * alphaX * alphaY * cos4Theta * Sqr(1 + e));
    }
    double Lambda(V3 w) const {
        double tan2Theta = Tan2Theta(w);
        if (std::isinf(tan2Theta)) return 0;
        double alpha2 = Sqr(CosPhi(w) * alphaX) + Sqr(SinPhi(w) * alphaY);
        return (std::sqrt(1 + alpha2 * tan2Theta) - 1) / 2;
    }
    double G1(V3 w) const { return 1 / (1 + Lambda(w)); }
    double G(V3 wo, V3 wi) const { return 1 / (1 + Lambda(wo) + Lambda(wi)); }
    double Dv(V3 w, V3 wm) const { return G1(w) / AbsCosTheta(w) * D(wm) * AbsDot(w, wm); } // D(w,wm) = PDF(w,wm)
    V3 SampleWm(V3 w, V2 u) const { // Material.h:412-435
        V3 wh = normalize(V3{alphaX * w.x, alphaY * w.y, w.z});
        if (wh.z < 0) wh = -wh;
        V3 T1 = (wh.z < 0.99999) ? normalize(cross(V3{0., 0., 1.}, wh)) : V3{1, 0, 0};
        V3 T2 = cross(wh, T1);
        V2 p = SampleUniformDiskPolar(u);
        double h = std::sqrt(1 - Sqr(p.x));
        p.y = Lerp((1 + wh.z) / 2, h, p.y);
        double pz = std::sqrt(std::max<double>(0., 1. - (Sqr(p.x) + Sqr(p.y))));
        V3 nh = p.x * T1 + p.y * T2 + pz * wh;
        return normalize(V3{alphaX * nh.x, alphaY * nh.y, std::max<double>(1e-6, nh.z)});
    }
    V3 Fresnel(V3 wo, V3 wm) const {
        return {FrComplex(AbsDot(wo, wm), Complex(eta.x, k.x)), FrComplex(AbsDot(wo, wm), Complex(eta.y, k.y)),
                FrComplex(AbsDot(wo, wm), Complex(eta.z, k.z))};
    }

    // ---- Sample
    MaterialSampleContext Sample(const MaterialEvalContext& c) const {
        MaterialSampleContext s;
        switch (type) {
        case MAT_LAMBERTIAN: { // Material.h:106-122
            V3 wi = SampleCosineHemisphere();
            while (wi.z <= 0.) wi = SampleCosineHemisphere();
            s.wi = wi;
            s.pdf = wi.z * InvPi;
            s.f = KdValue(c) * InvPi;
            s.flags = 1;
            return s;
        }
        case MAT_PHONG: { // Material.h:183-226
            double u = RandomDouble();
            if (u < pkd) {
                V3 wi = SampleCosineHemisphere();
                while (wi.z <= 0.) wi = SampleCosineHemisphere();
                s.wi = wi;
                s.pdf = wi.z * InvPi;
                s.f = KdValue(c) * InvPi;
                s.flags = 1;
            } else if (pkd <= u && u < pkd + pks) {
                double u1 = RandomDouble(), u2 = RandomDouble();
                double alpha = std::acos(std::pow(u1, 1.0 / (Ns + 1.0)));
                double phi = 2.0 * Pi * u2;
                double sinAlpha = std::sin(alpha), cosAlpha = std::cos(alpha), sinPhi = std::sin(phi),
                       cosPhi = std::cos(phi);
                V3 reflectWi{sinAlpha * cosPhi, sinAlpha * sinPhi, cosAlpha};
                V3 wi = ReflectiveSpaceToLocal(reflectWi, c);
                s.wi = wi;
                s.pdf = SpecularPDF(s.wi, c);
                V3 localReflect = normalize(Reflect(c.wo, V3{0., 0., 1.}));
                double localCosAlpha = std::max(0.0, dot(s.wi, localReflect));
                if (wi.z > 0. && localCosAlpha > 0.) {
                    s.f = KsValue(c) * (Ns + 2.) * Inv2Pi * std::pow(localCosAlpha, Ns);
                }
                s.flags = 2;
            }
            return s;
        }
        case MAT_MIRROR: { // Material.h:334-343
            s.wi = Reflect(c.wo, V3{0., 0., 1.});
            double cosTheta = s.wi.z;
            s.f = V3{1.0, 1.0, 1.0} / cosTheta;
            s.pdf = 1;
            return s;
        }
        case MAT_COOKTORRANCE: { // Material.h:437-472
            V3 wo = c.wo;
            if (wo.z == 0) return MaterialSampleContext{};
            double first = RandomDouble();
            double second = RandomDouble();
            V2 u{second, first}; // departure B20
            V3 wm = SampleWm(wo, u);
            V3 wi = Reflect(wo, wm);
            if (!SameHemisphere(wo, wi)) return MaterialSampleContext{};
            double pdf = Dv(wo, wm) / (4. * AbsDot(wo, wm));
            double cosTheta_o = AbsCosTheta(wo), cosTheta_i = AbsCosTheta(wi);
            if (cosTheta_i == 0 || cosTheta_o == 0) return MaterialSampleContext{};
            V3 F = Fresnel(wo, wm);
            V3 f = D(wm) * F * G(wo, wi) / (4. * cosTheta_i * cosTheta_o);
            s.f = f;
            s.pdf = pdf;
            s.wm = wm;
            s.wi = wi;
            s.flags = 4;
            return s;
        }
        default: return s; // Material.h:60-62
        }
    }

    // ---- Eval (NEE)
    V3 Eval(V3 wi, const MaterialEvalContext& c) const {
        switch (type) {
        case MAT_LAMBERTIAN: return KdValue(c) * InvPi; // Material.h:128-130
        case MAT_PHONG: {                              // Material.h:227-248 (draws one uniform)
            double u = RandomDouble();
            if (u < pkd) {
                if (wi.z <= 0) return {0, 0, 0};
                return KdValue(c) * InvPi;
            } else if (pkd <= u && u < pkd + pks) {
                if (wi.z <= 0) return {0, 0, 0};
                V3 localReflect = normalize(Reflect(c.wo, V3{0., 0., 1.}));
                double localCosAlpha = std::max(0., dot(wi, localReflect));
                if (localCosAlpha <= 0.) return {0, 0, 0};
                return KsValue(c) * (Ns + 2.) * Inv2Pi * std::pow(localCosAlpha, Ns);
            }
            return {0, 0, 0};
        }
        case MAT_COOKTORRANCE: { // Material.h:474-496
            V3 wo = c.wo;
            if (!SameHemisphere(wo, wi)) return {0, 0, 0};
            double cosTheta_o = AbsCosTheta(wo), cosTheta_i = AbsCosTheta(wi);
            if (cosTheta_i == 0 || cosTheta_o == 0) return {0, 0, 0};
            V3 wm = wi + wo;
            if (LengthSquared(wm) == 0) return {0, 0, 0};
            wm = normalize(wm);
            V3 F = Fresnel(wo, wm);
            return D(wm) * F * G(wo, wi) / (4 * cosTheta_i * cosTheta_o);
        }
        default: return {0, 0, 0}; // Material.h:63-65
        }
    }

    // ---- Scatter
    bool Scatter(const Ray& rayIn, const HitRecord& rec, V3& attenuation, Ray& scattered) const {
        MaterialEvalContext c;
        c.p = rec.position;
        c.uv = rec.uv;
        c.n = rec.normal;
        c.dpdus = rec.tangent;
        switch (type) {
        case MAT_LAMBERTIAN: { // Material.h:131-151
            c.wo = WorldToLocal(-rayIn.direction, rec);
            MaterialSampleContext s = Sample(c);
            scattered = Ray{rec.position, LocalToWorld(s.wi, c)};
            attenuation = s.f * s.wi.z / s.pdf;
            return true;
        }
        case MAT_PHONG: { // Material.h:263-285
            c.wo = WorldToLocal(-rayIn.direction, rec);
            MaterialSampleContext s = Sample(c);
            scattered = Ray{rec.position, LocalToWorld(s.wi, c)};
            if (s.pdf > 0. && s.wi.z > 0) attenuation = s.f * s.wi.z / s.pdf;
            else attenuation = V3{0, 0, 0}; // departure B13
            return true;
        }
        case MAT_MIRROR: { // Material.h:344-363
            c.wo = WorldToLocal(-rayIn.direction, rec);
            MaterialSampleContext s = Sample(c);
            scattered = Ray{rec.position, LocalToWorld(s.wi, c)};
            attenuation = s.f * s.wi.z / s.pdf;
            return true;
        }
        case MAT_COOKTORRANCE: { // Material.h:497-516
            c.wo = normalize(WorldToLocal(-rayIn.direction, rec));
            MaterialSampleContext s = Sample(c);
            if (s.flags == 0) return false;
            attenuation = s.f * s.wi.z / s.pdf;
            scattered = Ray{rec.position, LocalToWorld(s.wi, c)};
            return true;
        }
        default: return false; // Material.h:57-59
        }
    }
};

// ------------------------------------------------------------------ Triangle (Triangle.h, Triangle.cpp)
struct Triangle {
    V3 v[3], e[2];
    V2 uv[3];
    V3 normal, tangent;
    double area;
    AABB bbox;
    int material;
    double D;
    V3 w;
    int prim;

    void Init(const V3 vert[3], const V3 nrm[3], const V2 tc[3], int mat, int primId) { // Triangle.cpp:11-53
        for (int i = 0; i < 3; ++i) { v[i] = vert[i]; uv[i] = tc[i]; }
        material = mat;
        prim = primId;
        e[0] = v[1] - v[0];
        e[1] = v[2] - v[0];
        V3 n = cross(e[0], e[1]);
        normal = normalize(n);
        if (anynan(normal)) {
            normal = normalize(nrm[0] + nrm[1] + nrm[2]);
            if (anynan(normal)) normal = V3{0.0, 0.0, 1.0};
        }
        V2 d0 = uv[1] - uv[0];
        V2 d1 = uv[2] - uv[0];
        double f = 1.0 / (d0.x * d1.y - d1.x * d0.y);
        tangent.x = f * (d1.y * e[0].x - d0.y * e[1].x);
        tangent.y = f * (d1.y * e[0].y - d0.y * e[1].y);
        tangent.z = f * (d1.y * e[0].z - d0.y * e[1].z);
        tangent = normalize(tangent);
        if (anynan(tangent)) { // Triangle.cpp:40-46; 0.9f is a float literal compared against a double
            V3 vv = normal;
            V3 helper = (std::fabs(vv.x) < (double)0.9f) ? V3{1, 0, 0} : V3{0, 1, 0};
            tangent = normalize(cross(vv, helper));
        }
        area = length(n) * 0.5;
        D = dot(normal, v[0]);
        w = n / dot(n, n);
        AABB b0(v[0], v[1]); // Triangle.cpp:94-99
        AABB b1(v[0], v[2]);
        bbox = AABB(b0, b1);
    }
    bool Hit(const Ray& ray, Interval domain, HitRecord& rec) const { // Triangle.cpp:54-83, 100-113
        double denom = dot(normal, ray.direction);
        if (std::fabs(denom) < 1e-8) return false;
        double t = (D - dot(normal, ray.origin)) / denom;
        if (!domain.Contains(t)) return false;
        V3 p = ray.at(t);
        V3 v0p = p - v[0];
        double alpha = dot(w, cross(v0p, e[1]));
        double beta = dot(w, cross(e[0], v0p));
        if (alpha != alpha || beta != beta) return false;
        if ((alpha < 0) || (beta < 0) || (alpha + beta > 1)) return false;
        rec.uv = (1. - alpha - beta) * uv[0] + alpha * uv[1] + beta * uv[2];
        rec.position = p;
        rec.time = t;
        rec.material = material;
        rec.tangent = tangent;
        rec.SetFaceNormal(ray, normal);
        rec.prim = prim;
        rec.alpha = alpha;
        rec.beta = beta;
        return true;
    }
    void Sample(V3 origin, HitRecord& rec, double& pdf) const { // Triangle.cpp:84-93
        double x = std::sqrt(RandomDouble()), y = RandomDouble();
        V3 p = v[0] * (1.0 - x) + v[1] * (x * (1.0 - y)) + v[2] * (x * y);
        rec.position = p;
        V3 direction = p - origin;
        rec.SetFaceNormal(Ray{origin, direction}, normal);
        rec.material = material;
        rec.prim = prim;
        pdf = 1.0 / area;
    }
};

// ------------------------------------------------------------------ BVHNode (BVH.h, BVH.cpp)
// A Hittable reference: kind 0 = Triangle, 1 = BVHNode.
struct Ref {
    int kind, idx;
};

struct Scene;
struct BVHNode {
    AABB bbox;
    Ref left, right;
    double area = 0.0;
};

struct Scene {
    std::vector<Triangle> tris;
    std::vector<Material> mats;
    std::vector<Texture> texs;
    std::vector<BVHNode> nodes;
    Ref world{1, -1};  // world = HittableList(BVHNode(world))   main.cpp:44
    Ref lights{1, -1}; // lights = HittableList(BVHNode(lights)) main.cpp:45
    bool hasLights = false;

    const AABB& Box(Ref r) const { return r.kind == 0 ? tris[r.idx].bbox : nodes[r.idx].bbox; }
    double Area(Ref r) const { return r.kind == 0 ? tris[r.idx].area : nodes[r.idx].area; }

    // BVHNode::BVHNode(objects, start, end), BVH.cpp:7-48.  Sorts `objects` in place like the reference.
    int Build(std::vector<Ref>& objects, size_t start, size_t end) {
        BVHNode node;
        node.bbox = EmptyBox();
        for (size_t i = start; i < end; ++i) node.bbox = AABB(node.bbox, Box(objects[i]));
        int axis = node.bbox.LongestAxis();
        size_t span = end - start;
        if (span == 1) {
            node.left = node.right = objects[start];
            node.area = Area(objects[start]);
        } else if (span == 2) {
            node.left = objects[start];
            node.right = objects[start + 1];
            node.area = Area(objects[start]) + Area(objects[start + 1]);
        } else {
            std::sort(objects.begin() + start, objects.begin() + end,
                      [&](const Ref& a, const Ref& b) { return Box(a).Axis(axis).min < Box(b).Axis(axis).min; });
            size_t mid = start + span / 2;
            int l = Build(objects, start, mid);
            int r = Build(objects, mid, end);
            node.left = Ref{1, l};
            node.right = Ref{1, r};
            node.area = nodes[l].area + nodes[r].area;
        }
        nodes.push_back(node);
        return (int)nodes.size() - 1;
    }

    bool Hit(Ref r, const Ray& ray, Interval domain, HitRecord& rec) const {
        if (r.kind == 0) return tris[r.idx].Hit(ray, domain, rec);
        const BVHNode& n = nodes[r.idx]; // BVH.cpp:51-61
        if (!n.bbox.Hit(ray, domain)) return false;
        bool hl = Hit(n.left, ray, domain, rec);
        bool hr = Hit(n.right, ray, Interval(domain.min, hl ? rec.time : domain.max), rec);
        return hl || hr;
    }
    // HittableList::Hit with its single child (HittableList.h:26-39)
    bool WorldHit(const Ray& ray, Interval domain, HitRecord& rec) const {
        HitRecord temp;
        if (Hit(world, ray, Interval(domain.min, domain.max), temp)) {
            rec = temp;
            return true;
        }
        return false;
    }

    // BVH.cpp:86-100 — `p` is a float
    void TraverseSample(V3 origin, Ref node, float p, HitRecord& rec, double& pdf) const {
        if (node.kind == 0) {
            tris[node.idx].Sample(origin, rec, pdf);
            pdf *= tris[node.idx].area;
            return;
        }
        const BVHNode& n = nodes[node.idx];
        if (p < Area(n.left)) TraverseSample(origin, n.left, p, rec, pdf);
        else TraverseSample(origin, n.right, p - Area(n.left), rec, pdf);
    }
    // lights.Sample: HittableList::Sample (HittableList.h:44-59) -> BVHNode::Sample (BVH.cpp:62-67)
    void LightsSample(V3 origin, HitRecord& rec, double& pdf) const {
        double areaSum = Area(lights);
        double p0 = RandomDouble() * areaSum; // consumed; one child => always picked (p0 <= areaSum)
        (void)p0;
        double p = std::sqrt(RandomDouble()) * Area(lights);
        TraverseSample(origin, lights, (float)p, rec, pdf);
        pdf /= Area(lights);
    }
    void LightOrder(Ref r, std::vector<int>& out) const {
        if (r.kind == 0) { out.push_back(tris[r.idx].prim); return; }
        const BVHNode& n = nodes[r.idx];
        LightOrder(n.left, out);
        if (!(n.right.kind == n.left.kind && n.right.idx == n.left.idx)) LightOrder(n.right, out);
    }
};

// ------------------------------------------------------------------ Camera (Camera.h, Camera.cpp)
struct Camera {
    int imageWidth, imageHeight, samplesPerPixel, maxDepth;
    V3 background;
    double fovy;
    V3 eye, lookAt, up;
    bool bSampleLights;
    double russianRoulette;
    V3 center, pixel00Location, pixelDeltaU, pixelDeltaV, u, v, w;
    double pixelSamplesScale;

    void Initialize() { // Camera.cpp:75-106
        imageWidth = (imageWidth < 1) ? 1 : imageWidth;
        imageHeight = (imageHeight < 1) ? 1 : imageHeight;
        double aspectRatio = double(imageWidth) / double(imageHeight);
        pixelSamplesScale = 1.0 / samplesPerPixel;
        center = eye;
        double focalLength = length(eye - lookAt);
        double theta = fovy * 0.01745329251994329576923690768489; // glm::radians
        double h = std::tan(theta / 2.0);
        double viewportHeight = 2. * h * focalLength;
        double viewportWidth = viewportHeight * aspectRatio;
        w = normalize(eye - lookAt);
        u = normalize(cross(up, w));
        v = cross(w, u);
        V3 viewportU = viewportWidth * u;
        V3 viewportV = viewportHeight * -v;
        pixelDeltaU = viewportU / (double)imageWidth;
        pixelDeltaV = viewportV / (double)imageHeight;
        V3 viewportUpperLeft = center - (focalLength * w) - viewportU / 2. - viewportV / 2.;
        pixel00Location = viewportUpperLeft + 0.5 * (pixelDeltaU + pixelDeltaV);
    }
    bool bPixelJitter = false;
    // Camera.cpp:110-111 (commented out upstream): offset = SampleSquare() = dvec2(xi-0.5, xi-0.5)
    // (RandomNumberGenerator.h:65-68); g++ evaluates the argument list right to left, so offset.y takes
    // the first draw.  Upstream would call this once per pixel; here the offset is drawn per sample.
    Ray GetRayJittered(int i, int j) const {
        double oy = g_rng.next() - 0.5;
        double ox = g_rng.next() - 0.5;
        V3 pixelSample = pixel00Location + ((double)i + ox) * pixelDeltaU + ((double)j + oy) * pixelDeltaV;
        return Ray{center, pixelSample - center};
    }
    Ray GetRay(int i, int j) const { // Camera.cpp:108-117
        V3 pixelSample = pixel00Location + ((double)i) * pixelDeltaU + ((double)j) * pixelDeltaV;
        return Ray{center, pixelSample - center};
    }
};

struct Tracer {
    const Scene& sc;
    const Camera& cam;
    bool reusePeek;
    OrcCounters cnt{0, 0, 0, 0};
    // Optional path signature of the sample being traced (oracle.h, ORC_TRACE_*): vertex v = maxDepth - depth.
    int32_t* trace = nullptr;
    bool mute = false; // below a continuation of zero throughput (the product does not trace those paths)
    void TraceVertex(int depth, int prim) {
        const int v = cam.maxDepth - depth;
        if (!trace || mute || v >= ORC_TRACE_VERTS) return;
        trace[0] = v + 1;
        trace[1 + 2 * v] = prim;
        trace[2 + 2 * v] = 0;
    }
    void TraceFlag(int depth, int bit) {
        const int v = cam.maxDepth - depth;
        if (trace && !mute && v < ORC_TRACE_VERTS) trace[2 + 2 * v] |= bit;
    }

    // Camera::RayColor, Camera.cpp:119-204.  `pre` = the peek hit of the caller (same ray, same interval).
    V3 RayColor(const Ray& ray, int depth, const HitRecord* pre) {
        if (depth < 0.) return {0., 0., 0.};
        HitRecord record;
        bool hit;
        if (pre) {
            record = *pre;
            hit = true;
        } else {
            cnt.hit_calls++;
            hit = sc.WorldHit(ray, Interval(0.0001, kInf), record);
        }
        if (!hit) {
            TraceVertex(depth, -1);
            return cam.background;
        }
        TraceVertex(depth, record.prim);
        const Material& mat = sc.mats[record.material];
        if (mat.HasEmission()) return mat.GetEmission();
        const V3 ps = record.position;
        V3 direct{0., 0., 0.}, scatter{0., 0., 0.};

        if (cam.bSampleLights && sc.hasLights && !mat.SkipLightSampling()) {
            double pdfLights = 0.0;
            HitRecord lrec;
            sc.LightsSample(ps, lrec, pdfLights);
            const V3 pl = lrec.position;
            V3 lightDirection = normalize(pl - ps);
            V3 lightNormal = lrec.normal;
            const Material& lightMaterial = sc.mats[lrec.material];
            HitRecord shadowRec;
            double distance = length(pl - ps);
            Ray shadowRay{ps, lightDirection};
            cnt.hit_calls++;
            cnt.rays_shadow++;
            bool shadowHit = sc.WorldHit(shadowRay, Interval(0.001, std::numeric_limits<double>::max()), shadowRec);
            // departure B9: an escaping shadow ray counts as unoccluded
            bool visible = !shadowHit || (distance - length(ps - shadowRec.position) < 0.001);
            if (dot(record.normal, lightDirection) > 0.0 && lrec.bFrontFace) TraceFlag(depth, ORC_TRACE_NEE);
            if (dot(record.normal, lightDirection) > 0.0 && lrec.bFrontFace && visible) {
                TraceFlag(depth, ORC_TRACE_VISIBLE);
                V3 emission = lightMaterial.GetEmission();
                MaterialEvalContext context;
                context.p = record.position;
                context.uv = record.uv;
                context.n = record.normal;
                context.dpdus = record.tangent;
                context.wo = Material::WorldToLocal(-ray.direction, record);
                V3 localWi = Material::WorldToLocal(lightDirection, record);
                V3 localLightNormal = Material::WorldToLocal(lightNormal, record);
                V3 fr = mat.Eval(localWi, context);
                double cosTheta = localWi.z;
                double cosThetaBar = dot(localLightNormal, -localWi);
                direct = emission * fr * cosTheta * cosThetaBar / (distance * distance) / pdfLights;
            }
        }

        Ray scatteredRay{{0, 0, 0}, {0, 0, 0}};
        V3 attenuation{0, 0, 0};
        if (RandomDouble() < cam.russianRoulette) {
            TraceFlag(depth, ORC_TRACE_ROULETTE);
            if (mat.Scatter(ray, record, attenuation, scatteredRay)) {
                TraceFlag(depth, ORC_TRACE_SCATTER);
                const bool wasMute = mute;
                if (attenuation.x == 0. && attenuation.y == 0. && attenuation.z == 0.) mute = true;
                if (cam.bSampleLights) {
                    HitRecord peek;
                    cnt.hit_calls++;
                    cnt.rays_closest++;
                    if (sc.WorldHit(scatteredRay, Interval(0.0001, kInf), peek)) {
                        const HitRecord* pass = reusePeek ? &peek : nullptr;
                        if (!sc.mats[peek.material].HasEmission()) {
                            scatter = attenuation * RayColor(scatteredRay, depth - 1, pass) / cam.russianRoulette;
                        } else if (mat.SkipLightSampling()) {
                            scatter = attenuation * RayColor(scatteredRay, depth - 1, pass) / cam.russianRoulette;
                        } else if (depth - 1 >= 0) {
                            TraceVertex(depth - 1, peek.prim); // the vertex exists (a light reached by a bounce): it just adds nothing (Camera.cpp:191-195)
                        }
                    } else if (depth - 1 >= 0) {
                        TraceVertex(depth - 1, -1); // bounce miss: adds nothing (Camera.cpp:187)
                    }
                } else {
                    if (depth - 1 >= 0) cnt.rays_closest++;
                    scatter = attenuation * RayColor(scatteredRay, depth - 1, nullptr) / cam.russianRoulette;
                }
                mute = wasMute;
            }
        }
        return direct + scatter;
    }

    V3 Sample(int i, int j, int s, uint64_t seed) {
        g_rng.seed(seed, (uint64_t)j * (uint64_t)cam.imageWidth + (uint64_t)i, (uint64_t)s);
        Ray ray = cam.bPixelJitter ? cam.GetRayJittered(i, j) : cam.GetRay(i, j);
        cnt.samples++;
        cnt.rays_closest++;
        return RayColor(ray, cam.maxDepth, nullptr);
    }
};

Camera MakeCamera(const OrcCamera* c, const OrcRenderParams* p) {
    Camera cam;
    cam.imageWidth = c->width;
    cam.imageHeight = c->height;
    cam.samplesPerPixel = p->spp;
    cam.maxDepth = p->max_depth;
    cam.background = V3{p->background[0], p->background[1], p->background[2]};
    cam.fovy = c->fovy;
    cam.eye = V3{c->eye[0], c->eye[1], c->eye[2]};
    cam.lookAt = V3{c->look_at[0], c->look_at[1], c->look_at[2]};
    cam.up = V3{c->up[0], c->up[1], c->up[2]};
    cam.bSampleLights = p->sample_lights != 0;
    cam.russianRoulette = p->russian_roulette;
    cam.bPixelJitter = p->pixel_jitter != 0;
    cam.Initialize();
    return cam;
}

} // namespace

struct OrcScene {
    Scene sc;
};

extern "C" {

OrcScene* orc_scene_create(const OrcSceneDesc* d) {
    OrcScene* h = new OrcScene();
    Scene& sc = h->sc;
    sc.texs.resize(d->n_textures);
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const OrcTexture& t = d->textures[i];
        sc.texs[i].width = t.width;
        sc.texs[i].height = t.height;
        sc.texs[i].channels = t.channels;
        if (t.data) sc.texs[i].data.assign(t.data, t.data + (size_t)t.width * t.height * t.channels);
    }
    sc.mats.resize(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const OrcMaterial& m = d->materials[i];
        Material& o = sc.mats[i];
        o.type = m.type;
        o.tex = (m.texture >= 0) ? &sc.texs[m.texture] : nullptr;
        o.Kd = V3{m.kd[0], m.kd[1], m.kd[2]};
        o.Ks = V3{m.ks[0], m.ks[1], m.ks[2]};
        o.Ns = m.ns;
        o.SetProbabilitiesByNs();
        o.emission = V3{m.emission[0], m.emission[1], m.emission[2]};
        o.eta = V3{m.eta[0], m.eta[1], m.eta[2]};
        o.k = V3{m.k[0], m.k[1], m.k[2]};
        o.alphaX = m.alpha_x;
        o.alphaY = m.alpha_y;
    }
    sc.tris.resize(d->n_tris);
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
        for (uint64_t t = d->mesh_first_tri[m]; t < d->mesh_first_tri[m + 1]; ++t) {
            V3 v[3], n[3];
            V2 uv[3];
            for (int k = 0; k < 3; ++k) {
                v[k] = V3{d->vertices[t * 9 + k * 3], d->vertices[t * 9 + k * 3 + 1], d->vertices[t * 9 + k * 3 + 2]};
                n[k] = d->normals ? V3{d->normals[t * 9 + k * 3], d->normals[t * 9 + k * 3 + 1], d->normals[t * 9 + k * 3 + 2]}
                                  : V3{0, 0, 0};
                uv[k] = d->texcoords ? V2{d->texcoords[t * 6 + k * 2], d->texcoords[t * 6 + k * 2 + 1]} : V2{0, 0};
            }
            sc.tris[t].Init(v, n, uv, d->mesh_material[m], (int)t);
        }
    }
    // main.cpp:36-45 — per-mesh BVHNode into world (and, if emissive, a second BVHNode over the
    // already re-ordered mesh list into lights), then one top-level BVHNode over each list.
    std::vector<Ref> worldList, lightList;
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
        std::vector<Ref> objs;
        for (uint64_t t = d->mesh_first_tri[m]; t < d->mesh_first_tri[m + 1]; ++t) objs.push_back(Ref{0, (int)t});
        if (objs.empty()) continue;
        worldList.push_back(Ref{1, sc.Build(objs, 0, objs.size())});
        // main.cpp:40-41; with an explicit lights list (Camera::Render's second argument) the list decides instead
        if (!d->light_meshes && sc.mats[d->mesh_material[m]].HasEmission()) lightList.push_back(Ref{1, sc.Build(objs, 0, objs.size())});
    }
    if (d->light_meshes)
        for (uint32_t k = 0; k < d->n_light_meshes; ++k) {
            const uint32_t m = (uint32_t)d->light_meshes[k];
            std::vector<Ref> objs;
            for (uint64_t t = d->mesh_first_tri[m]; t < d->mesh_first_tri[m + 1]; ++t) objs.push_back(Ref{0, (int)t});
            if (objs.empty()) continue;
            sc.Build(objs, 0, objs.size()); // the world's BVHNode(mesh) sorted the mesh's list first (BVH.cpp:33)
            lightList.push_back(Ref{1, sc.Build(objs, 0, objs.size())});
        }
    if (!worldList.empty()) sc.world = Ref{1, sc.Build(worldList, 0, worldList.size())};
    if (!lightList.empty()) {
        sc.lights = Ref{1, sc.Build(lightList, 0, lightList.size())};
        sc.hasLights = true;
    }
    return h;
}

void orc_scene_destroy(OrcScene* h) { delete h; }

uint64_t orc_light_count(const OrcScene* h) {
    if (!h->sc.hasLights) return 0;
    std::vector<int> o;
    h->sc.LightOrder(h->sc.lights, o);
    return o.size();
}
void orc_light_order(const OrcScene* h, int32_t* prims) {
    if (!h->sc.hasLights) return;
    std::vector<int> o;
    h->sc.LightOrder(h->sc.lights, o);
    for (size_t i = 0; i < o.size(); ++i) prims[i] = o[i];
}

void orc_trace_closest(const OrcScene* h, const OrcRay* rays, size_t n, OrcHit* hits) {
    const Scene& sc = h->sc;
    for (size_t i = 0; i < n; ++i) {
        Ray r{{rays[i].o[0], rays[i].o[1], rays[i].o[2]}, {rays[i].d[0], rays[i].d[1], rays[i].d[2]}};
        HitRecord rec;
        OrcHit& o = hits[i];
        if (sc.world.idx >= 0 && sc.WorldHit(r, Interval(rays[i].tmin, rays[i].tmax), rec)) {
            o.t = rec.time;
            o.alpha = rec.alpha;
            o.beta = rec.beta;
            o.prim = rec.prim;
            o.front = rec.bFrontFace ? 1 : 0;
        } else {
            o.t = kInf;
            o.alpha = o.beta = 0;
            o.prim = -1;
            o.front = 0;
        }
    }
}

void orc_sample_lights(const OrcScene* h, const double* origins, size_t n, uint64_t seed, OrcLightSample* out) {
    const Scene& sc = h->sc;
    for (size_t i = 0; i < n; ++i) {
        g_rng.seed(seed, i, 0);
        HitRecord rec;
        double pdf = 0;
        sc.LightsSample(V3{origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]}, rec, pdf);
        OrcLightSample& o = out[i];
        o.position[0] = rec.position.x; o.position[1] = rec.position.y; o.position[2] = rec.position.z;
        o.normal[0] = rec.normal.x; o.normal[1] = rec.normal.y; o.normal[2] = rec.normal.z;
        o.pdf = pdf;
        o.prim = rec.prim;
        o.front = rec.bFrontFace ? 1 : 0;
    }
}

void orc_rng_stream(uint64_t seed, uint64_t pixel, uint64_t sample, size_t n, double* out) {
    Rng r;
    r.seed(seed, pixel, sample);
    for (size_t i = 0; i < n; ++i) out[i] = r.next();
}

void orc_render(const OrcScene* h, const OrcCamera* c, const OrcRenderParams* p, double* rgb, int y0, int y1,
                int nthreads, int reuse_peek, OrcCounters* counters) {
    Camera cam = MakeCamera(c, p);
    const Scene& sc = h->sc;
    if (nthreads < 1) nthreads = 1;
    if (y0 < 0) y0 = 0;
    if (y1 > cam.imageHeight) y1 = cam.imageHeight;
    std::vector<OrcCounters> cnts(nthreads, OrcCounters{0, 0, 0, 0});
    auto work = [&](int tid) {
        Tracer tr{sc, cam, reuse_peek != 0};
        for (int j = y0 + tid; j < y1; j += nthreads) { // Camera.cpp:50-58 (row bands -> interleaved rows)
            for (int i = 0; i < cam.imageWidth; ++i) {
                V3 acc{0, 0, 0};
                for (int s = 0; s < cam.samplesPerPixel; ++s) {
                    acc = acc + tr.Sample(i, j, s, p->seed) * cam.pixelSamplesScale; // Camera.cpp:56
                }
                size_t m = (size_t)j * cam.imageWidth + i;
                rgb[m * 3] = acc.x; rgb[m * 3 + 1] = acc.y; rgb[m * 3 + 2] = acc.z;
            }
        }
        cnts[tid] = tr.cnt;
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (counters) {
        OrcCounters tot{0, 0, 0, 0};
        for (auto& x : cnts) {
            tot.rays_closest += x.rays_closest; tot.rays_shadow += x.rays_shadow;
            tot.hit_calls += x.hit_calls; tot.samples += x.samples;
        }
        *counters = tot;
    }
}

void orc_render_samples(const OrcScene* h, const OrcCamera* c, const OrcRenderParams* p, const int32_t* pixel_xy,
                        size_t n_pixels, double* out) {
    orc_render_samples_trace(h, c, p, pixel_xy, n_pixels, 0, p->spp, out, nullptr);
}

void orc_render_samples_trace(const OrcScene* h, const OrcCamera* c, const OrcRenderParams* p, const int32_t* pixel_xy,
                              size_t n_pixels, int32_t sample_begin, int32_t sample_count, double* out, int32_t* trace) {
    Camera cam = MakeCamera(c, p);
    Tracer tr{h->sc, cam, true};
    for (size_t k = 0; k < n_pixels; ++k) {
        for (int s = 0; s < sample_count; ++s) {
            const size_t o = k * (size_t)sample_count + (size_t)s;
            if (trace) {
                tr.trace = trace + o * ORC_TRACE_WORDS;
                std::memset(tr.trace, 0, ORC_TRACE_WORDS * sizeof(int32_t));
            }
            V3 r = tr.Sample(pixel_xy[k * 2], pixel_xy[k * 2 + 1], sample_begin + s, p->seed);
            out[o * 3] = r.x; out[o * 3 + 1] = r.y; out[o * 3 + 2] = r.z;
        }
    }
}

// ---- material / texture hooks for the known-answer tests (same shapes as prt_material_* / prt_texture_value)
void orc_material_eval(const OrcScene* h, int32_t material, size_t n, const double* wi, const double* wo, const double* uv,
                       uint64_t seed, double* f) {
    const Material& m = h->sc.mats[material];
    for (size_t i = 0; i < n; ++i) {
        g_rng.seed(seed, i, 0);
        MaterialEvalContext c;
        c.p = V3{0, 0, 0};
        c.uv = uv ? V2{uv[i * 2], uv[i * 2 + 1]} : V2{0, 0};
        c.wo = V3{wo[i * 3], wo[i * 3 + 1], wo[i * 3 + 2]};
        c.n = V3{0, 0, 1};
        c.dpdus = V3{1, 0, 0};
        V3 r = m.Eval(V3{wi[i * 3], wi[i * 3 + 1], wi[i * 3 + 2]}, c);
        f[i * 3] = r.x; f[i * 3 + 1] = r.y; f[i * 3 + 2] = r.z;
    }
}

void orc_material_scatter(const OrcScene* h, int32_t material, size_t n, const double* rd, const double* normal,
                          const double* tangent, const double* uv, uint64_t seed, double* wi_world, double* attenuation,
                          int32_t* ok) {
    const Material& m = h->sc.mats[material];
    for (size_t i = 0; i < n; ++i) {
        g_rng.seed(seed, i, 0);
        HitRecord rec;
        rec.position = V3{0, 0, 0};
        rec.normal = V3{normal[0], normal[1], normal[2]};
        rec.tangent = V3{tangent[0], tangent[1], tangent[2]};
        rec.uv = uv ? V2{uv[i * 2], uv[i * 2 + 1]} : V2{0, 0};
        Ray in{V3{0, 0, 0}, V3{rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2]}};
        V3 att{0, 0, 0};
        Ray out{V3{0, 0, 0}, V3{0, 0, 0}};
        const bool good = m.Scatter(in, rec, att, out);
        ok[i] = good ? 1 : 0;
        wi_world[i * 3] = good ? out.direction.x : 0; wi_world[i * 3 + 1] = good ? out.direction.y : 0; wi_world[i * 3 + 2] = good ? out.direction.z : 0;
        attenuation[i * 3] = good ? att.x : 0; attenuation[i * 3 + 1] = good ? att.y : 0; attenuation[i * 3 + 2] = good ? att.z : 0;
    }
}

void orc_texture_value(const OrcScene* h, int32_t texture, size_t n, const double* uv, double* rgb) {
    const Texture& t = h->sc.texs[texture];
    for (size_t i = 0; i < n; ++i) {
        V3 c = t.Value(uv[i * 2], uv[i * 2 + 1]);
        rgb[i * 3] = c.x; rgb[i * 3 + 1] = c.y; rgb[i * 3 + 2] = c.z;
    }
}

// CookTorrance building blocks of one material: out[i] = {D(wm), Lambda(w), G1(w), D(w, wm)} and the conductor
// Fresnel term per channel for cos_i = |w . wm| (Material.h:373-411, MaterialUtils.h:100-111)
void orc_cooktorrance_terms(const OrcScene* h, int32_t material, size_t n, const double* w, const double* wm, double* out4,
                            double* fresnel3) {
    const Material& m = h->sc.mats[material];
    for (size_t i = 0; i < n; ++i) {
        const V3 a{w[i * 3], w[i * 3 + 1], w[i * 3 + 2]}, b{wm[i * 3], wm[i * 3 + 1], wm[i * 3 + 2]};
        out4[i * 4] = m.D(b);
        out4[i * 4 + 1] = m.Lambda(a);
        out4[i * 4 + 2] = m.G1(a);
        out4[i * 4 + 3] = m.Dv(a, b);
        const V3 F = m.Fresnel(a, b);
        fresnel3[i * 3] = F.x; fresnel3[i * 3 + 1] = F.y; fresnel3[i * 3 + 2] = F.z;
    }
}

// Material::WorldToLocal (to_local != 0) / LocalToWorld (Material.h:76-98) in the frame (normal, tangent)
void orc_frame(const double* normal, const double* tangent, size_t n, const double* in, int to_local, double* out) {
    HitRecord rec;
    rec.normal = V3{normal[0], normal[1], normal[2]};
    rec.tangent = V3{tangent[0], tangent[1], tangent[2]};
    MaterialEvalContext c;
    c.n = rec.normal;
    c.dpdus = rec.tangent;
    for (size_t i = 0; i < n; ++i) {
        const V3 v{in[i * 3], in[i * 3 + 1], in[i * 3 + 2]};
        const V3 r = to_local ? Material::WorldToLocal(v, rec) : Material::LocalToWorld(v, c);
        out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z;
    }
}

void orc_camera_rays(const OrcCamera* c, double* out) {
    OrcRenderParams p{};
    p.spp = 1;
    Camera cam = MakeCamera(c, &p);
    for (int j = 0; j < cam.imageHeight; ++j)
        for (int i = 0; i < cam.imageWidth; ++i) {
            Ray r = cam.GetRay(i, j);
            double* o = out + ((size_t)j * cam.imageWidth + i) * 6;
            o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
            o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
        }
}

} // extern "C"
