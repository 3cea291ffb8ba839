// tools/lab/plan_lab.cpp -- the planner's partition (recursive bisection + the refinement along patch borders: tp_plan.h, pk_build_plan steps 1 and 1b)
// under weights supplied from outside, for tools/plan_lab.py: what would the patches' loads be if the planner weighed the vertices differently?
// g++ -O2 -std=c++17 -shared -fPIC tools/lab/plan_lab.cpp -o tests/_build/libplan_lab.so
#include "../../tpose_amd/csrc/tp_plan.h"
#include <map>
extern "C" int lab_partition(int NP, int NT, const int32_t* tris, const float* points, int W, int H, float ratio, const double* wv, int parts, int passes, int32_t* owner_out) {
    std::map<std::pair<int, int>, int> eid;
    std::vector<int32_t> eu;
    for (int t = 0; t < NT; t++)
        for (int k = 0; k < 3; k++) {
            const int o = tris[4 * t + k], d = tris[4 * t + (k + 1) % 3];
            const std::pair<int, int> key(o < d ? o : d, o < d ? d : o);
            if (eid.emplace(key, (int)eid.size()).second) { eu.push_back(key.first); eu.push_back(key.second); }
        }
    std::vector<int> deg(NP, 0);
    for (int t = 0; t < NT; t++) for (int k = 0; k < 3; k++) deg[tris[4 * t + k]]++;
    std::vector<pk_detail::rcb_vertex> a;
    for (int v = 0; v < NP; v++) {
        if (!deg[v]) continue;
        float x = points[2 * (size_t)v] / ratio * 0.5f * (float)W, y = points[2 * (size_t)v + 1] * 0.5f * (float)H;
        a.push_back({v, x, y, wv[v]});
    }
    std::vector<int32_t> owner((size_t)NP, -1);
    pk_detail::rcb(a, 0, (int)a.size(), 0, parts, owner);
    std::vector<double> load((size_t)parts, 0.0);
    std::vector<int> count((size_t)parts, 0);
    for (auto& q : a) { load[owner[q.v]] += q.w; count[owner[q.v]]++; }
    const int NE = (int)eu.size() / 2;
    for (int pass = 0; pass < passes; pass++) {
        int moved = 0;
        for (int e = 0; e < NE; e++) {
            int u = eu[2 * e], v = eu[2 * e + 1];
            int A = owner[u], B = owner[v];
            if (A == B) continue;
            if (load[A] < load[B]) { std::swap(u, v); std::swap(A, B); }
            if (count[A] <= 1 || load[A] - load[B] <= wv[u]) continue;
            owner[u] = B;
            load[A] -= wv[u]; load[B] += wv[u]; count[A]--; count[B]++;
            moved++;
        }
        if (!moved) break;
    }
    for (int v = 0; v < NP; v++) owner_out[v] = owner[v];
    return 0;
}
// the planner's own weights for the same mesh (speed: pixels per grad-iter and vertex, or null)
extern "C" int lab_vertex_work(int NP, int NT, const int32_t* tris, const float* points, int H, const float* speed, double* wv_out) {
    std::map<std::pair<int, int>, int> eid;
    std::vector<int32_t> edge_uv, he_edge(3 * (size_t)NT);
    for (int t = 0; t < NT; t++)
        for (int k = 0; k < 3; k++) {
            const int o = tris[4 * t + k], d = tris[4 * t + (k + 1) % 3];
            const std::pair<int, int> key(o < d ? o : d, o < d ? d : o);
            auto it = eid.find(key);
            if (it == eid.end()) { it = eid.emplace(key, (int)(edge_uv.size() / 2)).first; edge_uv.push_back(key.first); edge_uv.push_back(key.second); }
            he_edge[3 * t + k] = it->second * 2 + (o != key.first ? 1 : 0);
        }
    std::vector<float> rows; std::vector<double> wv; std::vector<int> deg;
    pk_vertex_work(NP, NT, tris, points, (int)(edge_uv.size() / 2), edge_uv.data(), he_edge.data(), H, speed, rows, wv, deg);
    for (int v = 0; v < NP; v++) wv_out[v] = wv[v];
    return 0;
}
