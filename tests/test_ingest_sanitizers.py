"""The native TCP front-end (csrc/ingest.cpp: accept, epoll receive, tick and send threads) under ThreadSanitizer and under
Address + UndefinedBehaviour sanitizers.  tests/native/ingest_san.cpp compiles the front-end into one program with stubs of
the engine entry points, opens it over a step function and drives it with in-process clients: steady traffic (every frame
answered by the right listener), senders that vanish in the middle of a frame and reconnect, listeners that never read, a
close under load.  Any sanitizer report fails the test.  (gcc's libtsan and libasan ship with the image; no GPU involved.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "ingest_san.cpp")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not on PATH")
@pytest.mark.parametrize("name,flags,markers", [
    ("thread", ["-fsanitize=thread"], ["WARNING: ThreadSanitizer"]),
    ("address+undefined", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], ["ERROR: AddressSanitizer", "runtime error:", "LeakSanitizer"]),
])
def test_front_end_under_sanitizer(tmp_path, name, flags, markers):
    exe = tmp_path / "ingest_san"
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-Wno-subobject-linkage"] + flags + ["-o", str(exe), SRC],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    log = r.stdout + r.stderr
    for m in markers:
        assert m not in log, f"{name} sanitizer report:\n" + log[-6000:]
    assert r.returncode == 0 and "ok: phase 1" in log, log[-3000:]
