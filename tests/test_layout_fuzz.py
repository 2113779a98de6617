"""CPU: the host-side layout builders under AddressSanitizer + UBSan on randomised matrices (sanitizers run on the
CPU build only; the GPU pool has none)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_builders_under_sanitizers(tmp_path):
    exe = str(tmp_path / "layout_fuzz")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-fno-omit-frame-pointer", "-pthread", os.path.join(ROOT, "tools", "layout_fuzz.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_layout_builder_threads_under_thread_sanitizer(tmp_path):
    """The builder classifies rows, runs both counting sorts and tiles the fragments on several host threads; the same
    randomised matrices under ThreadSanitizer (fewer trials: it is slow), layouts compared with the one-thread build."""
    exe = str(tmp_path / "layout_fuzz_tsan")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-pthread",
                    os.path.join(ROOT, "tools", "layout_fuzz.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, "10", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
