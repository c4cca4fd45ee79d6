"""ray-tracing_amd/csrc/RayTracer_mi355x.cpp defines the public methods of the reference's class RayTracer over the C
ABI.  Where the reference tree is mounted, oracle/Makefile compiles it against the reference's own headers together
with a caller in the shape of App::run (tests/ref_class_frame_loop.cpp); the GPU suite runs that program
(tests/test_gpu_cpp_host.py).  Here: the program exists after a build, and without a GPU it ends with the library's
message instead of crashing."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_class_frame_loop")


def test_backend_builds_against_the_reference_headers():
    if not os.path.isdir("/root/reference/src/rt"):
        pytest.skip("the reference tree is not mounted here")
    assert os.path.isfile(EXE), "run make -C oracle ref (or __graft_entry__.build())"


def test_without_a_gpu_the_program_reports_the_library_message():
    if not os.path.isfile(EXE):
        pytest.skip("oracle/_ref/ref_class_frame_loop is built where the reference's headers are")
    run = subprocess.run([EXE], capture_output=True, text=True, timeout=100)
    if run.returncode == 0:
        assert run.stdout.count("\n") == 4          # a GPU is here: the four stages
    else:
        assert run.returncode == 1 and run.stdout.startswith("error: rt_create")
