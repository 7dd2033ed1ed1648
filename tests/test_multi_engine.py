"""The multi-device engine behind vsc_multi_search / vsc_multi_search_stream (csrc/vsc_multi.cpp, as shipped) on the CPU: built
with a host stand-in of the device layer (tools/multi_tsan/stub_device.cpp: copies deferred to the stream synchronisation, an
invented but checkable shard result, a merge that verifies every record) and driven by tools/multi_tsan/driver.cpp - 1-7 shards,
plain and streamed, every scoring mode, ragged and many batches, a callback that stops the stream, a shard that fails.  What is
tested is the engine's PROTOCOL (threads, exchange slots, batch order, error paths), not a search: the records are made up.
tools/multi_tsan/run.sh runs the same program under ThreadSanitizer and ASan + UBSan (profiles/r04_multi_tsan.txt); the real
searches over several contexts are tests/test_gpu_parity.py's."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tools", "multi_tsan")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_multi_device_engine_protocol_on_a_host_stand_in(tmp_path):
    exe = tmp_path / "multi_engine"
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "varscot_amd", "csrc"), "-I" + HERE, os.path.join(ROOT, "varscot_amd", "csrc", "vsc_multi.cpp"),
           os.path.join(HERE, "stub_device.cpp"), os.path.join(HERE, "driver.cpp"), "-pthread", "-ldl", "-o", str(exe)]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "180 streamed runs over 1..7 shards" in r.stdout and ", 0 failures" in r.stdout
