// view -- headless counterpart of the reference's software/view program: the flat-shaded picture of one
// level of a stacked .tri file, drawn with the stored triangle colours at mix(points, originpoints, s)
// (software/view/main.cpp:95-120, shader/triangle.vs), written as a binary PPM instead of a window.
//
//   view -t file.tri [-level K] [-s 0..1] [-height 600] [-o out.ppm] [-device D]
//
// The reference opens a RATIO*600 x 600 window and lets s oscillate between 0 and 1; SPACE steps to the
// next level of the stack (here: -level K reads K+1 records).
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "tpose/io.hpp"
#include "tpose/triangulation.hpp"

using namespace tpose;

int main(int argc, char** argv) {
    std::string tri, out = "view.ppm";
    int level = 0, height = 600, device = 0;
    float s = 0.0f;
    for (int a = 1; a < argc; a++) {
        const std::string k = argv[a];
        auto val = [&]() -> const char* { if (a + 1 >= argc) { std::cerr << "missing value for " << k << "\n"; std::exit(2); } return argv[++a]; };
        if (k == "-t") tri = val();
        else if (k == "-level") level = std::atoi(val());
        else if (k == "-s") s = (float)std::atof(val());
        else if (k == "-height") height = std::atoi(val());
        else if (k == "-o") out = val();
        else if (k == "-device") device = std::atoi(val());
        else { std::cerr << "unknown option " << k << "\n"; return 2; }
    }
    if (tri.empty()) { std::cout << "Please specify a input triangulation with -t." << std::endl; return 0; }
    io::verbose = false;
    triangulation tr;
    for (int k = 0; k <= level; k++)
        if (!io::read(&tr, tri)) { std::cout << "no level " << k << " in " << tri << std::endl; return 0; }
    const int width = (int)(RATIO * (float)height);  // Tiny::window(..., tpose::RATIO*600, 600)
    tpose::init(width, height, device);
    tpose::upload(&tr);
    std::vector<uint8_t> rgba((size_t)width * height * 4);
    tpose::draw_stored(&tr, s, rgba.data(), (size_t)width * 4);
    tpose::quit();
    FILE* f = std::fopen(out.c_str(), "wb");
    if (!f) { std::cerr << "cannot write " << out << "\n"; return 1; }
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    for (size_t i = 0; i < (size_t)width * height; i++) std::fwrite(&rgba[4 * i], 1, 3, f);
    std::fclose(f);
    std::cout << "wrote " << out << " (" << width << "x" << height << ", NT=" << tr.NT << ")" << std::endl;
    return 0;
}
