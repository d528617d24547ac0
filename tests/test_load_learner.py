"""CPU: the load learner of round 6 (rolo_amd/csrc/load_learner.hpp — what a context learns from its own frames' device time about load its process cannot count) on synthetic
frame durations taken from the measured regimes of profiles/r06/load_regimes.json: alone it never leaves the idle-device kernels, beside a second process it finds and keeps the
busy-device ones and goes back when the process leaves, beside a copy stream it stays, and a try that does not pay is not repeated for 512 frames."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_load_learner_scenarios(tmp_path):
    exe = str(tmp_path / "learner_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "rolo_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "learner_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "all held" in r.stdout, r.stdout + r.stderr
