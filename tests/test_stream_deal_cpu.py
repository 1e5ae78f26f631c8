"""The stream deal of the frames-in-flight pipe (d2slam_amd/csrc/stream_deal.h, used by place_streams() in csrc/pipe.hip) compiled with g++ and run on the host:
the arrangements the pipe must produce from the class sequences measured on MI355X, and the invariants of any deal (tests/cpp/deal_test.cpp).  The measurement
itself (which streams take turns on the device) needs the GPU: tests/test_pipe.py::test_pipe_stream_placement_is_measured_and_separates_a_lanes_streams."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_deal_cases(tmp_path):
    exe = str(tmp_path / "deal_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "d2slam_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "deal_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
