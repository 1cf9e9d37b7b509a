"""The lock-free ready-queue protocol of forge_ec_amd/csrc/sched_lf.hpp, restated for host threads
(tests/cpp/sched_lf_model.cpp): twelve threads as the twelve wavefronts of a scheduler workgroup, the same counters,
atomics, entry format and batch policy; every element claimed once and stepped to its end, no slot in two batches, no
entry overwritten unread, every thread terminates -- under the OS's arbitrary descheduling, which is how the first form
of the protocol (no consumed flag) was caught.  Also checked: the constants the model restates are the header's."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "sched_lf_model.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "sched_lf_model")


def test_model_and_header_agree_on_the_protocol_constants():
    hdr = open(os.path.join(ROOT, "forge_ec_amd", "csrc", "sched_lf.hpp")).read()
    mdl = open(SRC).read()
    assert "constexpr int LF_BIAS = 1 << 16;" in hdr and "constexpr int64_t kBias = 1 << 16;" in mdl
    assert "constexpr unsigned LF_ERRFLAG = 1u << 30;" in hdr and "constexpr uint32_t kErrFlag = 1u << 30;" in mdl
    assert "constexpr u32 LF_CONSUMED = 0x400u;" in hdr and "0x400u | lap(" in mdl
    assert re.search(r"#define FEC_LF_TAIL_SHIFT 2\b", hdr) and "int th = (int)(remain >> 2);" in mdl
    # the lap tag: (pos / RING) mod 32 in bits 11..15, both ring sizes
    assert "return RING == 1024 ? ((pos << 1) & 0xF800u) : (pos & 0xF800u);" in hdr
    assert "return ring == 1024 ? ((pos << 1) & 0xF800u) : (pos & 0xF800u);" in mdl
    # the producer looks before it writes, the consumer marks what it has read
    assert "const u32 expect = LF_CONSUMED | lf_lap<RING>(pos - (u32)RING);" in hdr
    assert "if (active) lf_write_b16(entry, LF_CONSUMED | want_tag);" in hdr


def test_lock_free_queue_model_runs_clean():
    if not os.path.exists(EXE) or os.path.getmtime(SRC) > os.path.getmtime(EXE):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-Wextra", "-o", EXE, SRC])
    r = subprocess.run([EXE, "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "540 runs, every element claimed once" in r.stdout
