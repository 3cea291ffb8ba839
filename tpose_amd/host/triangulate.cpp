// triangulate -- headless counterpart of the reference's software/triangulate program: energy-based
// image triangulation with the reference's frame schedule (software/triangulate/main.cpp:190-353),
// driven through the tpose:: host mirror (include/tpose/) on top of the HIP C ABI.
//
//   triangulate -i image.ppm [-o out.tri] [-window 1.5] [-maxframes N] [-maxtris N] [-levels 50,100,...]
//               [-device D] [-quiet] [-literal]
//
// One frame = doenergy, doshift, read back tenergy/penergy/colnum/points, then -- once the relative
// energy change drops below 1e-4 -- export (on the 50,100,...,1000 ladder), energy-sorted flip set with
// flip-back, split of the worst triangle; every frame: prune, wide-angle flips, short-edge collapses;
// then computecolors at the new positions.  No window, no GL: `-window f` only selects the raster
// (image size / f, like the reference's Tiny::window(w/1.5, h/1.5)); default f = 1 (raster == image).
// `-literal`: the frame as the reference writes it -- all 13 NT entries of the buffers read back every frame, every angle and
// every shortest edge evaluated, a comparison sort for the flip ranking -- instead of the shortcuts that decide the same
// (tests compare the two byte for byte).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <map>
#include <set>
#include <string>
#include <algorithm>
#include <chrono>
#include <vector>

#include "tpose/io.hpp"
#include "tpose/triangulation.hpp"
#include "image_io.hpp"

using namespace tpose;

int main(int argc, char** argv) {
    std::string input, output;
    float window = 1.0f;
    long maxframes = 1L << 40;
    int maxtris = 1 << 30, device = 0;
    std::string levels;  // export list override (the reference hard-codes 50..1000; its showcase uses 3000)
    bool quiet = false, literal = false, nochunks = false;
    for (int a = 1; a < argc; a++) {
        const std::string k = argv[a];
        auto val = [&]() -> const char* { if (a + 1 >= argc) { std::cerr << "missing value for " << k << "\n"; std::exit(2); } return argv[++a]; };
        if (k == "-i") input = val();
        else if (k == "-o") output = val();
        else if (k == "-window") window = (float)std::atof(val());
        else if (k == "-maxframes") maxframes = std::atol(val());
        else if (k == "-maxtris") maxtris = std::atoi(val());
        else if (k == "-levels") levels = val();
        else if (k == "-device") device = std::atoi(val());
        else if (k == "-quiet") quiet = true;
        else if (k == "-literal") literal = true;
        else if (k == "-nochunks") nochunks = true;   // frame by frame with the shortcuts of round 5, no tp_iterate_frames
        else { std::cerr << "unknown option " << k << "\n"; return 2; }
    }
    if (input.empty()) { std::cout << "Please specify an input image with -i." << std::endl; return 0; }
    Raster img;
    if (!load_raster(input, img)) { std::cout << "Failed to load image." << std::endl; return 0; }
    if (output.empty()) output = input + ".tri";
    io::verbose = !quiet;

    std::vector<int> exportlist = {1000, 900, 800, 700, 600, 500, 400, 300, 200, 100, 50};  // consumed from the back
    if (!levels.empty()) {
        exportlist.clear();
        size_t pos = 0;
        while (pos < levels.size()) {
            const size_t comma = levels.find(',', pos);
            exportlist.insert(exportlist.begin(), std::atoi(levels.substr(pos, comma - pos).c_str()));
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
    }
    const int nlevels = (int)exportlist.size();
    const auto t_start = std::chrono::steady_clock::now();

    RATIO = (float)img.w / (float)img.h;
    Raster raster = img;
    if (window != 1.0f) raster = resample(img, (int)((float)img.w / window), (int)((float)img.h / window));

    tpose::init(raster.w, raster.h, device);
    tpose::flavour = TP_TRIANGULATE;
    tpose::image(TP_IMAGE_A, raster.rgba.data(), (size_t)raster.w * 4);

    triangulation tr;
    tpose::upload(&tr, false);
    if (!quiet) std::cout << "Number of Triangles: " << tr.NT << std::endl;
    tpose::computecolors();
    // The reference's frame is doenergy, doshift, four readbacks, topology work, computecolors (its render pipeline).
    // The trailing computecolors only feeds the next frame's doenergy, so a frame that follows one without topology
    // work runs computecolors + doenergy + doshift as one fused device sequence (tpose::doframe) -- same buffers.
    bool fresh = true;  // colacc / colnum belong to the current triangulation and positions

    long frame = 0;
    bool done = false;
    double t_device = 0, t_converged = 0, t_loops = 0, t_flip = 0, t_rank = 0, t_upload = 0, t_energy = 0, t_reup = 0;  // where the wall time goes (stderr, with "seconds")
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    // The next frame rides ahead (round 5): a frame whose energy has not converged is followed by the host's prune / wide-angle / collapse
    // sweeps over every triangle (17 us at 3000 triangles) and, four times in five, by nothing else -- the next frame is then the same fused
    // device sequence whatever the sweeps found NOT to do.  So it is enqueued BEFORE the sweeps and runs while the host sweeps; when a sweep
    // does change the mesh, the upload that follows overwrites everything that frame touched (positions, topology, colour sums), and the frame
    // is simply never read back.  Decisions, states and bytes are the reference's (`-literal` keeps its order of calls as written).
    bool ahead = false;   // the device is already running the frame the loop is about to count
    // the per-frame sweeps of software/triangulate/main.cpp:316-346 over tr as it stands (positions of the frame just read): true when the mesh
    // changed in a way the device has to hear of (prune, collapse; the wide-angle flips stay on the host, as in the reference)
    auto sweeps = [&]() {
        bool changed = false;
        for (size_t t = 0; t < (size_t)tr.NT; t++)
            if (tr.boundary((int)t) == 3)
                if (tr.prune((int)t)) changed = true;
        for (size_t t = 0; t < (size_t)tr.NT; t++) {
            int maybe = literal ? 7 : tr.sweep_candidates((int)t);   // (bit k: angle(3 t + k) may exceed 0.8 PI)
            for (int k = 0; k < 3; k++)
                if (((maybe >> k) & 1) && tr.angle(3 * (int)t + k) > 0.8 * tpose::PI) {
                    tr.flip(3 * (int)t + k, 0.0);
                    if (!literal) maybe = tr.sweep_candidates((int)t);   // (the flip may have changed triangle t)
                }
        }
        for (size_t t = 0; t < tr.triangles.size(); t++) {
            if (!literal && !(tr.sweep_candidates((int)t) & 8)) continue;   // (collapse() would refuse whichever half-edge is the shortest)
            int h = 3 * (int)t;
            float shortest = tr.hlength(h);
            if (tr.hlength(h + 1) < shortest) shortest = tr.hlength(++h);
            if (tr.hlength(h + 1) < shortest) shortest = tr.hlength(++h);
            if (tr.collapse(h)) changed = true;
        }
        return changed;
    };
    // Frames in chunks (round 6, tp_iterate_frames).  On a photograph the reference's convergence test holds a level for thousands of frames --
    // resource/meninas.png: ~10 000 frames per split below 50 triangles -- and every one of them was a device round trip (17 us) for a host
    // that looks, finds nothing to do and asks for the next.  After QUIET_MIN frames in a row without a convergence step or a change of the
    // mesh, frames run in chunks on the device and the host replays its part -- geterr, then the sweeps, frame by frame in the reference's
    // order -- over the energies and positions every frame left in the device's rings; the first frame that converges or changes the mesh
    // ends the run (the frames the device ran beyond it are dropped), and the loop goes on from exactly the state the frame-by-frame
    // loop would be in.  Decisions, frame counts and .tri bytes are those of `-literal` (tests/test_configs.py, tests/test_harness.py).
    const long QUIET_MIN = 16;
    long calm = 0, budget = 64;
    double t_chunks = 0; long chunk_frames = 0, chunk_calls = 0;
    while (!done && frame < maxframes) {
        bool have_frame = false, conv = false, swept = false, updated = false, device_stale = false;
        // (a chunk's first launch cuts a plan of the mesh -- about a microsecond per triangle on the host -- where a frame on its own costs 20: a
        // big mesh has to have been calm for long before its frames go in chunks)
        if (!literal && !nochunks && !ahead && !fresh && calm >= std::max<long>(QUIET_MIN, tr.NT / 4) && maxframes - frame >= 4) {
            const auto c0 = now();
            int kind = 0;   // 1: the last frame handed over converged (replayed on the device), 2: its sweeps changed the mesh
            const long n = tpose::frames(&tr, std::min(budget, maxframes - frame), [&](int) {
                frame++;
                if (tpose::geterr(&tr) < 1E-4) { kind = 1; return (int)TP_FRAME_STOP_REPLAY; }
                if (sweeps()) { kind = 2; return (int)TP_FRAME_STOP; }
                return (int)TP_FRAME_GO_ON;
            });
            t_chunks += secs(c0, now()); chunk_frames += n; chunk_calls++;
            if (kind == 0) { budget = std::min<long>(budget * 2, 4096); continue; }   // (nothing happened: the device holds the last frame's positions)
            calm = 0; budget = 64;
            have_frame = true;
            if (kind == 1) { conv = true; tpose::retrieve(&tr, true); }   // (the frame once more on the device: its buffers, as after doframe())
            else { swept = true; updated = true; }
        }
        const auto t0 = now();
        if (!have_frame) {
            frame++;
            if (!ahead) {
                if (fresh) { tpose::doenergy(); tpose::doshift(); }
                else tpose::doframe();
            }
            ahead = false;
            fresh = false;
            tpose::retrieve(&tr, !literal);   // (the reference reads all 13 NT entries of three buffers every frame and looks at the first NT)
            conv = tpose::geterr(&tr) < 1E-4;
        }
        const auto t1 = now();
        t_device += secs(t0, t1);
        if (conv) calm = 0; else calm++;

        if (conv) {
            if (exportlist.empty() || tr.NT > maxtris) { done = true; break; }
            if (tr.NT >= exportlist.back()) {
                tpose::retrieve_colors(&tr);
                for (int i = 0; i < tr.NT; i++)
                    if (tpose::cn[i] > 0) tr.colors[i] /= tpose::cn[i];
                tr.originpoints = tr.points;
                io::write(&tr, output);
                exportlist.pop_back();
            }

            // half-edges ordered by the energy of their triangle pair, highest first; of several with EQUAL energy
            // only the first one inserted survives -- the reference keeps them in a std::set keyed on the energy
            // alone.  Same order and same survivors from a stable sort + unique (a set of 3 NT nodes per
            // convergence step is the costliest host work of the schedule).
            const auto r0 = now();
            // (the order is that of a stable sort by descending energy: a stable radix sort on the float's bits, which order
            // like the values for the non-negative sums at hand; ties stay in insertion order, 3t + k ascending)
            // (... and the two half-edges of an edge carry the same energy, so the later one never survives the `unique` below: the
            // shortcut path does not insert it in the first place -- half the entries to sort; the buffers live across the steps)
            static std::vector<std::pair<int, float>> ranked, spare;
            static std::vector<uint32_t> count;
            ranked.clear();
            ranked.reserve(tr.triangles.size() * 3);
            bool radix_ok = !literal;
            for (int t = 0; t < (int)tr.triangles.size(); t++)
                for (int k = 0; k < 3; k++) {
                    const int w = tr.halfedges[3 * t + k];
                    if (w < 0) continue;
                    if (!literal && w < 3 * t + k) continue;
                    const float e = tpose::terr[t] + tpose::terr[w / 3];
                    radix_ok = radix_ok && e >= 0.0f;   // (false for negative sums, -0 and NaN alike: the comparison sort then)
                    ranked.emplace_back(3 * t + k, e == 0.0f ? 0.0f : e);
                }
            if (radix_ok) {
                spare.resize(ranked.size());
                for (int pass = 0; pass < 3; pass++) {   // digits of 11, 11 and 10 bits, least significant first, descending
                    const int shift = 11 * pass, bits = pass == 2 ? 10 : 11;
                    count.assign((size_t)1 << bits, 0);
                    auto digit = [&](float e) { uint32_t b; memcpy(&b, &e, 4); return ((~b) >> shift) & ((1u << bits) - 1u); };
                    for (auto& r : ranked) count[digit(r.second)]++;
                    uint32_t run = 0;
                    for (auto& c : count) { const uint32_t n = c; c = run; run += n; }
                    for (auto& r : ranked) spare[count[digit(r.second)]++] = r;
                    ranked.swap(spare);
                }
            } else {
                std::stable_sort(ranked.begin(), ranked.end(), [](const std::pair<int, float>& l, const std::pair<int, float>& r) { return l.second > r.second; });
            }
            ranked.erase(std::unique(ranked.begin(), ranked.end(), [](const std::pair<int, float>& l, const std::pair<int, float>& r) { return l.second == r.second; }), ranked.end());
            static std::vector<char> locked;                    // half-edges whose triangle already takes part in a flip
            locked.assign(tr.halfedges.size(), 0);
            static std::vector<std::pair<int, float>> chosen;   // half-edge -> pair energy before the flip, by half-edge (the reference's std::map order)
            chosen.clear();
            for (auto& h : ranked) {
                if (locked[h.first]) continue;
                const int w = tr.halfedges[h.first];
                if (w < 0) continue;
                if (locked[w]) continue;
                chosen.push_back(h);
                for (int k = 0; k < 3; k++) { locked[3 * (h.first / 3) + k] = 1; locked[3 * (w / 3) + k] = 1; }
            }
            std::sort(chosen.begin(), chosen.end(), [](const std::pair<int, float>& l, const std::pair<int, float>& r) { return l.first < r.first; });
            const auto r0b = now();
            for (auto& h : chosen) tr.flip(h.first, 0.0f);
            const auto r1 = now();
            t_flip += secs(r0b, r1);
            // The reference makes the flip set real to look at it: upload, computecolors, doenergy, read `tenergy` back -- and then at two
            // entries per flipped edge.  A triangle's energy depends on its own pixels only, so those are the base energies of the two
            // triangles the flip WOULD leave: evaluated on the device at its current positions (= tr.points: read back this frame), nothing
            // uploaded (`-literal` does what the reference does; tests compare the outputs byte for byte).
            bool evaluated = false;
            std::vector<int> pair_energy;   // [2 i], [2 i + 1]: the two triangles of chosen[i] after its flip
            auto r2 = now();                // (the evaluation counts as an energy pass in the timing summary, not as an upload)
            if (!literal) {
                std::vector<int> vs;
                vs.reserve(chosen.size() * 6);
                for (auto& h : chosen)
                    for (int t : {h.first / 3, tr.halfedges[h.first] / 3})
                        for (int k = 0; k < 3; k++) vs.push_back(tr.triangles[t][k]);
                evaluated = tpose::evaluate(vs, pair_energy);
            }
            if (!evaluated) {
                tpose::upload(&tr, false);
                r2 = now();
                tpose::computecolors();
                tpose::doenergy();
                tpose::retrieve_energy(&tr, !literal);
            }
            const auto r3 = now();
            for (size_t i = 0; i < chosen.size(); i++) {   // undo the flips that raised their pair's energy
                const auto& h = chosen[i];
                const int after = evaluated ? pair_energy[2 * i] + pair_energy[2 * i + 1] : tpose::terr[h.first / 3] + tpose::terr[tr.halfedges[h.first] / 3];
                if (after > h.second) tr.flip(h.first, 0.0f);
            }
            const auto r4 = now();
            t_flip += secs(r3, r4);
            // ... and the energies of the mesh the flips left -- the reference's second "upload, computecolors, doenergy, read back": the same
            // NT + 2 entries from the device without making the mesh real there.  The device then holds the mesh of before the flips until
            // the upload at the end of this frame (a split follows almost always); if none comes, the mesh goes up there on its own.
            auto r5 = r4;
            if (evaluated && tpose::evaluate_mesh_energy(&tr)) device_stale = true;
            else {
                tpose::upload(&tr, false);
                r5 = now();
                tpose::computecolors();
                tpose::doenergy();
                tpose::retrieve_energy(&tr, !literal);
            }
            const auto r6 = now();
            t_rank += secs(r0, r1) + secs(r3, r4); t_upload += secs(r1, r2) + secs(r4, r5); t_energy += secs(r2, r3) + secs(r5, r6);

            const int worst = tpose::maxerrid(&tr);
            if (worst >= 0 && tr.split(worst)) updated = true;
        }
        else if (!literal && frame < maxframes && !swept && (nochunks || calm < std::max<long>(QUIET_MIN, tr.NT / 4))) { tpose::doframe(); ahead = true; }   // (no convergence step: the next frame, ahead of the sweeps -- unless the next frames run in a chunk)
        const auto t2 = now();
        t_converged += secs(t1, t2);

        if (!swept && sweeps()) updated = true;
        const auto t3 = now();
        if (updated) {
            calm = 0;
            const float e = tpose::gettoterr(&tr);
            if (!quiet) std::cout << tr.NT << " " << std::setprecision(16) << e << std::endl;
            tpose::upload(&tr, false);
            tpose::computecolors();  // a new topology: the sweep cannot ride the next frame's fused sequence
            fresh = true;            // (upload drops the device lists; keep the reference's order of calls)
            ahead = false;           // (the frame that rode ahead ran on the mesh of before: overwritten, never read back)
        } else if (device_stale) {
            tpose::upload(&tr, false);   // (flips the device has not seen, and nothing else changed: the next frame's fused sequence needs them)
        }
        device_stale = false;
        t_loops += secs(t2, t3);
        t_reup += secs(t3, now());
    }
    std::cout << "frames " << frame << " triangles " << tr.NT << " points " << tr.NP << " levels written "
              << (nlevels - (int)exportlist.size()) << std::endl;
    std::cerr << "frame device calls + readbacks " << t_device << " s, convergence steps (flip set, split) " << t_converged
              << " s, per-frame host loops (prune, angle, collapse) " << t_loops << " s, re-upload + computecolors after a change " << t_reup << " s" << std::endl;
    std::cerr << "inside the convergence steps: ranking + flips " << t_rank << " s (the flips and flip-backs themselves " << t_flip << "), uploads " << t_upload << " s, computecolors + doenergy + read-back "
              << t_energy << " s" << std::endl;
    std::cerr << "frames in chunks on the device (tp_iterate_frames) " << chunk_frames << " of " << frame << " in " << chunk_calls << " calls, " << t_chunks << " s" << std::endl;
    std::cerr << "seconds " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << std::endl;
    tpose::quit();
    return 0;
}
