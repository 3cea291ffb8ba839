// tpose/io.hpp -- stacked binary `.tri` files and the text match list, byte-compatible with the
// reference's tpose::io (source/io.hpp:20-220).
//
// One record per hierarchy level, appended back to back, native little-endian, no magic, no padding:
//     float RATIO | int NT | NT x { int v0,v1,v2 ; int h0,h1,h2 ; int r,g,b } | int NP | NP x { float px,py ; float ox,oy }
// read() keeps tri->in open and consumes ONE record per call (false + close at end of file);
// write() keeps tri->out open and appends ONE record per call.  read() overwrites tpose::RATIO.
// With dowarp the incoming level's points are pushed through the CURRENT level's warp before the
// triangulation is replaced (coarse-to-fine hand-down, software/warp/main.cpp:272-280).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "triangulation.hpp"

namespace tpose {
namespace io {

template <class T>
inline void get(std::ifstream& s, T& v) { s.read(reinterpret_cast<char*>(&v), sizeof(T)); }
template <class T>
inline void put(std::ofstream& s, const T& v) { s.write(reinterpret_cast<const char*>(&v), sizeof(T)); }

inline bool verbose = true;

// "xA yA xB yB" per line; lines that do not parse are skipped.  Like the reference's loop (source/io.hpp:20-60: getline,
// then break on eof before parsing), a last line WITHOUT a trailing newline is dropped.
inline bool readmatches(std::string file, std::vector<vec2>& A, std::vector<vec2>& B) {
    if (verbose) std::cout << "Importing from file " << file << std::endl;
    std::ifstream f(file, std::ios::in);
    if (!f.is_open()) {
        std::cout << "Failed to open file " << file << std::endl;
        return false;
    }
    std::string line;
    while (std::getline(f, line)) {
        if (f.eof()) break;  // the unterminated last line (getline hit end-of-file inside it)
        vec2 a, b;
        if (std::sscanf(line.c_str(), "%f %f %f %f", &a.x, &a.y, &b.x, &b.y) == 4) {
            A.push_back(a);
            B.push_back(b);
        }
    }
    return true;
}

inline bool read(tpose::triangulation* tri, std::string file, bool dowarp = false) {
    if (verbose) std::cout << "Importing triangulation from " << file << " ... ";
    if (!tri->in.is_open()) {
        tri->in.open(file, std::ios::binary | std::ios::in);
        if (!tri->in.is_open()) {
            std::cout << "failed to open file." << std::endl;
            std::exit(0);
        }
    }
    get(tri->in, tpose::RATIO);
    if (tri->in.eof()) {
        tri->in.close();
        if (verbose) std::cout << "end of file." << std::endl;
        return false;
    }
    get(tri->in, tri->NT);
    std::vector<ivec4> tris(tri->NT), cols(tri->NT);
    std::vector<int> hes(3 * (size_t)tri->NT);
    for (int t = 0; t < tri->NT; t++) {
        for (int k = 0; k < 3; k++) get(tri->in, tris[t][k]);
        tris[t].w = 0;
        for (int k = 0; k < 3; k++) get(tri->in, hes[3 * t + k]);
        for (int k = 0; k < 3; k++) get(tri->in, cols[t][k]);
        cols[t].w = 1;
    }
    get(tri->in, tri->NP);
    std::vector<vec2> pts(tri->NP), origin(tri->NP);
    for (int p = 0; p < tri->NP; p++) {
        get(tri->in, pts[p].x); get(tri->in, pts[p].y);
        get(tri->in, origin[p].x); get(tri->in, origin[p].y);
    }
    if (dowarp) tri->warp(pts);  // through the level still held by *tri
    tri->triangles = tris;
    tri->colors = cols;
    tri->halfedges = hes;
    tri->points = pts;
    tri->originpoints = origin;
    if (verbose) std::cout << "success (" << tri->NT << ")." << std::endl;
    return true;
}

inline void write(tpose::triangulation* tri, std::string file) {
    if (!tri->out.is_open()) {
        tri->out.open(file, std::ios::binary | std::ios::out);
        if (!tri->out.is_open()) {
            std::cout << "Failed to open file " << file << std::endl;
            std::exit(0);
        }
    }
    if (verbose) std::cout << "Exporting to " << file << std::endl;
    put(tri->out, tpose::RATIO);
    put(tri->out, tri->NT);
    for (int t = 0; t < tri->NT; t++) {
        for (int k = 0; k < 3; k++) put(tri->out, tri->triangles[t][k]);
        for (int k = 0; k < 3; k++) put(tri->out, tri->halfedges[3 * t + k]);
        for (int k = 0; k < 3; k++) put(tri->out, tri->colors[t][k]);
    }
    put(tri->out, tri->NP);
    for (int p = 0; p < tri->NP; p++) {
        put(tri->out, tri->points[p].x); put(tri->out, tri->points[p].y);
        put(tri->out, tri->originpoints[p].x); put(tri->out, tri->originpoints[p].y);
    }
    tri->out.flush();
}

}  // namespace io
}  // namespace tpose
