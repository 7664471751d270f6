"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over what runs on the CPU (SURVEY.md §5 "race detection / sanitizers"; GPU sanitizers are
not available on this pool): (1) the oracle — everything the parity tests trust — driven at small, ragged, degenerate and failing shapes
through every family of its entry points (oracle/asan_driver.c, f32 and f64 builds); (2) the C ABI's host-side logic — validation of a
problem description, the weight count, the step-record layout arithmetic, the option block, grid checks (csrc/lde_host.h, which
lde_api.hip is built from) — with 200 000 hostile descriptions (tests/host_logic_driver.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
           OMP_NUM_THREADS="2")


def _clean(r):
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    for bad in ("AddressSanitizer", "runtime error:", "LeakSanitizer"):
        assert bad not in r.stderr and bad not in r.stdout, r.stderr[-3000:]


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_oracle_under_asan_ubsan(prec):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", f"oracle_asan_{prec}")], capture_output=True, text=True, env=ENV, timeout=600)
    _clean(r)
    assert "rc = 0" in r.stdout


def test_c_abi_host_logic_under_asan_ubsan(tmp_path):
    cxx = shutil.which("g++")
    exe = os.path.join(tmp_path, "host_logic_driver")
    subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-Wall", "-Wextra", "-Wno-unused-function", "-o", exe, os.path.join(ROOT, "tests", "host_logic_driver.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    _clean(r)
    assert "accepted" in r.stdout and "forward mappings as measured" in r.stdout


def test_lde_api_is_built_from_the_checked_logic():
    """lde_api.hip must USE lde_host.h's functions (not keep private copies that the sanitizers never see)."""
    src = open(os.path.join(ROOT, "latentdiffeq.jl_amd", "csrc", "lde_api.hip")).read()
    assert '#include "lde_host.h"' in src and "using namespace lde_host" in src
    pend = open(os.path.join(ROOT, "latentdiffeq.jl_amd", "csrc", "lde_pendulum.hip")).read()
    assert "lde_host::pend_forward_mapping(" in pend and "tn.sh_max_b" not in pend and "tn.tl_max_b" not in pend   # the launch code follows the checked function, no thresholds of its own
    for fn in ("static int validate(", "static size_t rec_bytes(", "static lde::StepRec rec_view(", "static lde::KOpts make_opts("):
        assert fn not in src, fn
