"""GPU: the C ABI from a C++ caller (tests/stubs/host_paths_check.cpp) - window views, multi-pass launches, the re-queue
variants, inclusion subsets, streamed handles (from arrays and from BLAST v4 volumes), ranges of a volume, end points,
the alignment phase with a too-small text buffer, both strands / two queries per pass, translated shards from arrays
and from volumes - checked by self-consistency (every option variant reproduces the default scores, every hit list is
the ordered top of those scores, a streamed or file-backed handle answers like the resident one).  The same program,
statically linked with the host objects built under ASan + UBSan (`make -C swipe_amd/csrc asan`, tools/asan_cli.sh),
is how the host side of these paths runs instrumented on the device (profiles/r03_asan.txt)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_abi_host_paths_are_self_consistent_from_a_cpp_caller(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "host_paths_check")
    lib = os.path.join(ROOT, "swipe_amd")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "stubs", "host_paths_check.cpp"),
                            "-L" + lib, "-lswipe_amd", "-Wl,-rpath," + lib], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    r = subprocess.run([exe, "0", "7", "2", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ", 0 bad" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
