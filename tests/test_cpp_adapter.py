"""The C++ adapter (hip_trace_backend.hpp) over the C ABI: compiles with plain g++, links the in-tree library,
throws BackendUnavailableError without a GPU and runs a session with one."""
import os
import subprocess

import pytest

from ice_halo_sim_amd import backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_main")


def _build():
    backend.load_library()
    libdir = os.path.join(ROOT, "ice_halo_sim_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", EXE, os.path.join(ROOT, "tests", "cpp", "adapter_main.cpp"),
                           "-L" + libdir, "-lhalo_hip", "-ldl", "-Wl,-rpath," + libdir])


def test_adapter_compiles_and_reports_unavailable_without_gpu():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    if backend.load_library().halo_device_count() == 0:
        assert r.returncode == 3, r.stdout + r.stderr
        assert "BackendUnavailableError" in r.stdout


@pytest.mark.gpu
def test_adapter_runs_a_session_on_the_gpu():
    _build()   # always: a binary that travelled with the snapshot may predate the header
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)   # (a hang must fail this test, not stall the suite)
    assert r.returncode == 0, r.stdout + r.stderr
    # the native multi-GPU drain (halo_reduce_accumulator: lazy dlopen of librccl, ncclReduce on the backend's stream) must have RUN,
    # on a one-rank communicator, and left the root's image as it was
    assert "rccl one-rank reduce:" in r.stdout and "image_unchanged 1" in r.stdout, r.stdout
