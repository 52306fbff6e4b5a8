// Small vector / matrix value types with the names and memory layouts the particle solver's plugin
// surface uses (the reference's catpaw/geometry.h:30-60, 160, 194, 297, 519-533 under
// /root/reference/particle_solver): cfloat3 is three tightly packed floats, cfloat4 four.  Only what
// the solver host code and main.cpp need is here; this is not a general maths library.
#pragma once

#include <cmath>

struct cfloat2 {
    float x = 0.f, y = 0.f;
    cfloat2() = default;
    cfloat2(float x_, float y_) : x(x_), y(y_) {}
};

struct cfloat3 {
    float x, y, z;
    cfloat3() : x(0.f), y(0.f), z(0.f) {}
    cfloat3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    cfloat3 operator+(const cfloat3 &o) const { return cfloat3(x + o.x, y + o.y, z + o.z); }
    cfloat3 operator-(const cfloat3 &o) const { return cfloat3(x - o.x, y - o.y, z - o.z); }
    cfloat3 operator*(float s) const { return cfloat3(x * s, y * s, z * s); }
    cfloat3 operator/(float s) const { return cfloat3(x / s, y / s, z / s); }
    cfloat3 &operator+=(const cfloat3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    cfloat3 &operator-=(const cfloat3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    cfloat3 &operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    float Norm() const { return std::sqrt(x * x + y * y + z * z); }
};
static_assert(sizeof(cfloat3) == 12, "cfloat3 must be three packed floats");

inline float dot(const cfloat3 &a, const cfloat3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct cint3 {
    int x, y, z;
    cint3() : x(0), y(0), z(0) {}
    cint3(int x_, int y_, int z_) : x(x_), y(y_), z(z_) {}
    int prod() const { return x * y * z; }
};

struct cfloat4 {
    float x, y, z, w;
    cfloat4() : x(0.f), y(0.f), z(0.f), w(0.f) {}
    cfloat4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};
static_assert(sizeof(cfloat4) == 16, "cfloat4 must be four packed floats");

// row-major square matrices; the particle step never touches them, main.cpp only names the types
struct cmat2 { float data[4] = {0.f, 0.f, 0.f, 0.f}; };
struct cmat3 { float data[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; };
struct cmat4 { float data[16] = {0.f}; };
class GivensRotation;

// cloth connectivity of one particle (indices of the neighbouring lattice nodes, -1 = none)
struct adjacent {
    int up = -1, down = -1, left = -1, right = -1;
    int upleft = -1, upright = -1, downleft = -1, downright = -1;
    int upup = -1, downdown = -1, leftleft = -1, rightright = -1;
};

namespace catpaw {}
