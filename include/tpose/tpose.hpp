// tpose/tpose.hpp -- library-wide state of the tpose host mirror.
// Mirrors source/tpose.hpp:12 of the reference: the aspect ratio of the domain is a process-global
// that every triangulation consults (x in [-RATIO, RATIO], y in [-1, 1], y up) and that io::read
// overwrites (source/io.hpp:81).  The reference's TinyEngine instancing models (source/tpose.hpp:26-42)
// have no counterpart: the HIP path needs no vertex models.
#pragma once

namespace tpose {
inline float RATIO = 12.0f / 8.0f;
}  // namespace tpose
