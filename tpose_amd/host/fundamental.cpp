// fundamental -- headless counterpart of the reference's tests/compute_fundamental_mat tool (BASELINE
// config 5): warped-vertex correspondences -> fundamental matrix -> corrected matches / 3D points.
//
//   fundamental A.tri A.tri.warp B.tri B.tri.warp [-level K] [-all | -select t0,t1,...] [-points out.txt]
//   fundamental -matches data.txt [-image WxH]          (text matches "xA yA xB yB", tests/sfm_match_test;
//                                                        -image maps pixel matches into the t-pose domain, which
//                                                        the boundary filter and thresholds of F_RANSAC assume)
//
// The reference is an ImGui program: the user clicks triangles, "Compute F" gathers the vertices of the
// selected triangles of A (originpoints -> points) and of B (points -> originpoints), maps them to image
// coordinates with T (tests/compute_fundamental_mat/main.cpp:137-166) and prints F_Sampson / F_LMEDS /
// F_RANSAC.  Here the selection is a command-line list (default: every triangle), the same matrices are
// printed, and the mean squared Sampson distance of each is printed next to it (the figure BASELINE config 5
// compares by, since the reference's OpenCV RANSAC is unseeded).
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "tpose/io.hpp"
#include "tpose/multiview.hpp"
#include "tpose/triangulation.hpp"

using namespace tpose;

static void report(const char* name, const mview::Matrix3f& F, const std::vector<vec2>& X, const std::vector<vec2>& Y) {
    std::cout << name << ": " << F << std::endl;
    std::printf("%s mean squared Sampson distance: %.6e\n", name, mview::mean_sampson(F, X, Y));
}

int main(int argc, char** argv) {
    std::vector<std::string> files;
    std::string matches, points_out, select, dump;
    int level = 0, imw = 0, imh = 0;
    for (int a = 1; a < argc; a++) {
        const std::string k = argv[a];
        auto val = [&]() -> const char* { if (a + 1 >= argc) { std::cerr << "missing value for " << k << "\n"; std::exit(2); } return argv[++a]; };
        if (k == "-matches") matches = val();
        else if (k == "-level") level = std::atoi(val());
        else if (k == "-image") { if (std::sscanf(val(), "%dx%d", &imw, &imh) != 2) { std::cerr << "-image WxH\n"; return 2; } }
        else if (k == "-select") select = val();
        else if (k == "-all") select.clear();
        else if (k == "-points") points_out = val();
        else if (k == "-dumpmatches") dump = val();   // the matches the estimators get, "xA yA xB yB" per line, and F_Sampson
        else files.push_back(k);
    }
    io::verbose = false;
    std::vector<vec2> matchX, matchY;
    if (!matches.empty()) {
        if (!io::readmatches(matches, matchX, matchY)) return 1;
        if (imw > 0 && imh > 0) {
            tpose::RATIO = (float)imw / (float)imh;
            for (auto* v : {&matchX, &matchY})
                for (auto& p : *v) p = vec2((2.0f * p.x / (float)imw - 1.0f) * tpose::RATIO, 1.0f - 2.0f * p.y / (float)imh);
        }
    } else {
        if (files.size() < 4) { std::cout << "This needs at least 4 triangulation files (A, A.warp, B, B.warp)" << std::endl; return 0; }
        triangulation trA, trWA, trB, trWB;
        for (int k = 0; k <= level; k++) {
            if (!io::read(&trA, files[0]) || !io::read(&trWA, files[1]) || !io::read(&trB, files[2]) || !io::read(&trWB, files[3])) {
                std::cout << "no level " << k << std::endl; return 0;
            }
        }
        tpose::RATIO = 9.6f / 5.4f;  // the tool's fixed window (main.cpp:30); io::read overwrote it per file
        // the warped positions; origin points stay (main.cpp:60-68)
        trA.points = trWA.points;
        trB.points = trWB.points;
        std::vector<int> selA(trA.NT, select.empty() ? 1 : 0), selB(trB.NT, select.empty() ? 1 : 0);
        if (!select.empty()) {
            std::stringstream ss(select);
            std::string tok;
            while (std::getline(ss, tok, ',')) { const int t = std::atoi(tok.c_str()); if (t >= 0 && t < trA.NT) selA[t] = 1; if (t >= 0 && t < trB.NT) selB[t] = 1; }
        }
        std::vector<int> pointsA(trA.points.size(), 0), pointsB(trB.points.size(), 0);
        for (int i = 0; i < trA.NT; i++) if (selA[i]) { pointsA[trA.triangles[i].x] = pointsA[trA.triangles[i].y] = pointsA[trA.triangles[i].z] = 1; }
        for (int i = 0; i < trB.NT; i++) if (selB[i]) { pointsB[trB.triangles[i].x] = pointsB[trB.triangles[i].y] = pointsB[trB.triangles[i].z] = 1; }
        // T (main.cpp:141-145) is written as glm::mat3(0.5/R, 0, 1,  0, -0.5/R, 1/R,  0, 0, 1): glm fills COLUMNS, so
        // T * (x, y, 1) = (0.5/R x, -0.5/R y, x + y/R + 1), and the vec2 the match vectors keep is (0.5/R x, -0.5/R y)
        // -- the translation the author meant never reaches the matches.  Mirrored as written.
        const float R = tpose::RATIO;
        auto map = [&](vec2 p) { return vec2(0.5f / R * p.x, -0.5f / R * p.y); };
        int NPA = 0, NPB = 0;
        for (size_t i = 0; i < pointsA.size(); i++) if (pointsA[i]) { NPA++; matchX.push_back(map(trA.originpoints[i])); matchY.push_back(map(trA.points[i])); }
        for (size_t i = 0; i < pointsB.size(); i++) if (pointsB[i]) { NPB++; matchX.push_back(map(trB.points[i])); matchY.push_back(map(trB.originpoints[i])); }
        std::cout << "Found A Matches: " << NPA << std::endl;
        std::cout << "Found B Matches: " << NPB << std::endl;
    }
    if (matchX.size() < 8) { std::cout << "Insufficient Matches..." << std::endl; return 0; }
    const mview::Matrix3f FS = mview::F_Sampson(matchX, matchY);
    report("F_Sampson", FS, matchX, matchY);
    if (!dump.empty()) {
        FILE* f = std::fopen(dump.c_str(), "w");
        if (!f) return 1;
        for (size_t i = 0; i < matchX.size(); i++) std::fprintf(f, "%.9g %.9g %.9g %.9g\n", matchX[i].x, matchX[i].y, matchY[i].x, matchY[i].y);
        std::fclose(f);
        f = std::fopen((dump + ".F").c_str(), "w");
        if (!f) return 1;
        for (int r = 0; r < 3; r++) std::fprintf(f, "%.9g %.9g %.9g\n", FS(r, 0), FS(r, 1), FS(r, 2));
        std::fclose(f);
    }
    report("F_LMEDS", mview::F_LMEDS(matchX, matchY), matchX, matchY);
    report("F_RANSAC", mview::F_RANSAC(matchX, matchY), matchX, matchY);
    if (!points_out.empty()) {  // optimal correction + structure with the tool's intrinsics
        const std::vector<mview::vec4> P = mview::triangulate(FS, mview::Camera(), matchX, matchY);
        FILE* f = std::fopen(points_out.c_str(), "w");
        if (!f) return 1;
        for (auto& p : P) std::fprintf(f, "%.6f %.6f %.6f\n", p.x, p.y, p.z);
        std::fclose(f);
        std::cout << "wrote " << P.size() << " points to " << points_out << std::endl;
    }
    return 0;
}
