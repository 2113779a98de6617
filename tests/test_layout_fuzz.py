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
