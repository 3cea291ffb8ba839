"""C++ host mirror (include/tpose/*.hpp): known-answer tests against the fixtures SURVEY.md section 8c
lists for the reference's host half (topology ops, warp, geterr, .tri format), and 150 randomised sequences of
split / flip / optimize / collapse / prune with vertex moves in between, whose half-edge invariants (mutual twins over
the same edge, ids in range, sizes; while nothing was removed or moved also: one triangle per directed edge and the
triangles tile the domain) are checked after every operation.  CPU only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_host_mirror_known_answers(tmp_path):
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    exe = os.path.join(HERE, "_build", "test_topology")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(HERE, "host", "test_topology.cpp"), "-o", exe])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror OK" in out.stdout
