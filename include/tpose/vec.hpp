// tpose/vec.hpp -- the handful of vector types the tpose data model is expressed in.
//
// The reference (weigert/t-pose) uses glm (`using namespace glm` in source/triangulation.hpp:19):
// vec2 = 2 x f32, ivec2 = 2 x i32, ivec4 = 4 x i32 -- these layouts are what the GPU buffers and the
// .tri files hold.  glm is not a dependency of this build; the types below have the same layout,
// member names (x,y,z,w + operator[]) and float32 arithmetic, so code written against the
// reference's `tpose::triangulation` members compiles unchanged.  Define TPOSE_USE_GLM before
// including to use the real glm types instead.
#pragma once

#ifdef TPOSE_USE_GLM
#include <glm/glm.hpp>
namespace tpose {
using glm::ivec2;
using glm::ivec4;
using glm::vec2;
using glm::vec3;
using glm::dot;
using glm::length;
}  // namespace tpose
#else

#include <cmath>
#include <cstdint>

namespace tpose {

struct vec2 {
    float x, y;
    vec2() : x(0), y(0) {}
    vec2(float s) : x(s), y(s) {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
    float& operator[](int i) { return i == 0 ? x : y; }
    const float& operator[](int i) const { return i == 0 ? x : y; }
    vec2& operator+=(vec2 b) { x += b.x; y += b.y; return *this; }
    vec2& operator-=(vec2 b) { x -= b.x; y -= b.y; return *this; }
};
inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
inline vec2 operator*(float s, vec2 a) { return vec2(s * a.x, s * a.y); }
inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
inline vec2 operator/(vec2 a, float s) { return vec2(a.x / s, a.y / s); }
inline bool operator==(vec2 a, vec2 b) { return a.x == b.x && a.y == b.y; }
inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
inline float length(vec2 a) { return std::sqrt(dot(a, a)); }

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

struct ivec2 {
    int32_t x, y;
    ivec2() : x(0), y(0) {}
    ivec2(int32_t x_, int32_t y_) : x(x_), y(y_) {}
};

struct ivec4 {
    int32_t x, y, z, w;
    ivec4() : x(0), y(0), z(0), w(0) {}
    ivec4(int32_t x_, int32_t y_, int32_t z_, int32_t w_) : x(x_), y(y_), z(z_), w(w_) {}
    int32_t& operator[](int i) { return (&x)[i]; }
    const int32_t& operator[](int i) const { return (&x)[i]; }
    ivec4& operator/=(int32_t d) { x /= d; y /= d; z /= d; w /= d; return *this; }
};
inline bool operator==(const ivec4& a, const ivec4& b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

static_assert(sizeof(vec2) == 8 && sizeof(ivec2) == 8 && sizeof(ivec4) == 16, "GPU buffer layouts");

}  // namespace tpose
#endif
