"""CPU: the library's threaded host code under ThreadSanitizer and AddressSanitizer.  csrc/launch_engine.h -- condition
variables, detached engine threads, argument arrays for 96 folds, the drain-by-chunk protocol that decides WHEN a fold may have its buffers
back -- is compiled against tests/host_mock/mock_hip.h (streams = FIFOs drained by worker threads, events = tickets, kernel launches = closures
in stream order that dereference the folds' buffers) and driven by tests/host_mock/engine_soak.cpp: host threads hand folds to the engines,
are woken, free their buffers AT ONCE and come back; launch classes mixed, some folds given up at their cap, the engines filled to the brim.
GPU sanitizers are not available on the pool (and the GPU soaks of tools/soak_batch.py cannot see a host race that did not fire)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "host_mock")


def build(tmp, flag, src_dir=MOCK):
    exe = os.path.join(tmp, "engine_soak_" + flag.split("=")[1])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", flag, "-o", exe, os.path.join(src_dir, "engine_soak.cpp"), "-pthread"])
    return exe


@pytest.mark.parametrize("flag,env", [("-fsanitize=thread", {"TSAN_OPTIONS": "report_thread_leaks=0 halt_on_error=1 exitcode=66"}),
                                      ("-fsanitize=address", {"ASAN_OPTIONS": "detect_leaks=0 exitcode=66"})])
@pytest.mark.parametrize("threads,jobs", [(16, 8), (230, 3)])     # 230 threads: more folds than the two engines' 2 x 96 argument slots
def test_launch_engine_under_sanitizers(tmp_path, flag, env, threads, jobs):
    exe = build(str(tmp_path), flag)
    r = subprocess.run([exe, str(threads), str(jobs)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert f"engine soak ok: {threads * jobs} folds" in r.stdout, r.stdout
    print("\n" + r.stdout.strip())


def test_the_harness_sees_a_fold_woken_too_early(tmp_path):
    """Teeth: with the drain step removed from a COPY of launch_engine.h (a finished fold is released although the next chunk, already
    enqueued, still names it) the same soak must die under AddressSanitizer with a use-after-free in a mock kernel."""
    src = tmp_path / "tests" / "host_mock"
    hdr = tmp_path / "trrosettax2-dynamics_amd" / "csrc"
    src.mkdir(parents=True); hdr.mkdir(parents=True)
    for f in ("mock_hip.h", "engine_soak.cpp"):
        shutil.copy(os.path.join(MOCK, f), src / f)
    h = open(os.path.join(ROOT, "trrosettax2-dynamics_amd", "csrc", "launch_engine.h")).read()
    good = "if (in_next) { j->state = 2; j->drain_chunk = k; E->draining.push_back(j); }\n          else { j->state = 3; woke = true; }"
    assert h.count(good) == 1
    (hdr / "launch_engine.h").write_text(h.replace(good, "{ j->state = 3; woke = true; (void)in_next; }"))
    exe = build(str(tmp_path), "-fsanitize=address", str(src))
    r = subprocess.run([exe, "32", "16"], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0 exitcode=66"), capture_output=True, text=True, timeout=600)
    # (whichever sees it first: AddressSanitizer in a mock kernel, the soak's own check of a woken fold, or the mock kernels' order check reading
    # the counters of a fold whose buffers have been freed and handed to the next one)
    assert r.returncode != 0 and any(t in r.stderr for t in ("heap-use-after-free", "woken before", "mock: step of evaluation")), (r.returncode, r.stderr[-1500:])
