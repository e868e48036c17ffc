"""CPU: the C-ABI library loads, exports every symbol include/pt_hip.h declares, and fails loudly
(no CPU fallback) when there is no GPU. No compute call is made here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT
import path_tracer_amd
from path_tracer_amd import host


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert len(syms) >= 19
    assert sorted(path_tracer_amd.SIGNATURES) == syms


def test_library_exports_every_declared_symbol():
    lib = path_tracer_amd.load_library()
    for s in declared_symbols():
        assert getattr(lib, s) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", path_tracer_amd.LIB_PATH]).decode()
    exported = set(re.findall(r" T (pt_[a-z_0-9]+)", out))
    assert exported == set(declared_symbols())


def test_library_is_built_for_gfx950_only():
    blob = open(path_tracer_amd.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"nvptx"):
        assert other not in blob


def test_host_library_loads():
    lib = host.load_host_library()
    for s in host.HOST_SIGNATURES:
        assert getattr(lib, s) is not None


def _gpu_present():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_failure_not_fallback():
    lib = path_tracer_amd.load_library()
    h = C.c_void_p()
    rc = lib.pt_create(0, C.byref(h))
    assert rc != 0 and not h.value
    msg = lib.pt_last_error(None).decode()
    assert "no CPU path" in msg or "device" in msg
    with pytest.raises(path_tracer_amd.PtError):
        path_tracer_amd.Device(0)
    # the device-backed class surface fails too, instead of computing on the host
    with pytest.raises(RuntimeError):
        host.sphere_is_hit((0, 0, -5), 1.0, (0, 0, 0), (0, 0, -1))


def test_null_handles_are_rejected():
    lib = path_tracer_amd.load_library()
    assert lib.pt_destroy(None) != 0
    assert lib.pt_synchronize(None) != 0
    assert lib.pt_render(None, 8, 8, 0, 1, 5, 1, None, None) != 0
    assert lib.pt_create(0, None) != 0


def test_product_never_touches_the_oracle():
    """Nothing under path_tracer_amd/ or include/ may name the oracle."""
    for base in ("path_tracer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_compiled_kernels_pass_the_static_checks():
    """`make check` (part of `make all`): pt_device.h issues s_load_dwordx16 from inline asm and waits for it in a separate
    statement, so nothing between a load and its wait may read or write the registers being loaded (tools/check_sload.py,
    default flags and the PT_STATS build); and k_render<0|1|2> and k_tail must use no scratch memory, spill no vector
    register and keep their occupancy (tools/check_resources.py)."""
    import shutil
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    subprocess.check_call(["make", "-s", "-C", ROOT, "check", "HIPCC=" + hipcc], stdout=subprocess.DEVNULL)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_resources
    import check_sload
    n, problems = check_sload.check(os.path.join(ROOT, "tools", "build", "pt_hip.s"))
    assert n >= 4 and not problems, problems
    n, problems = check_sload.check(os.path.join(ROOT, "tools", "build", "pt_hip_stats.s"))
    assert n >= 4 and not problems, problems
    n, problems = check_resources.check(os.path.join(ROOT, "tools", "build", "pt_hip.resources.txt"))
    assert n >= 4 and not problems, problems
