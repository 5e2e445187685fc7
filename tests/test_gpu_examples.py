"""The drivers under examples/ — the reference's examples/cfd set-ups on this backend — run end to end at small sizes (each one checks
its own result with asserts and exits non-zero otherwise)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "flow_past_sphere_3d_hip.py": ["--nx", "96", "--ny", "48", "--nz", "48", "--steps", "300"],
    "rotating_sphere_3d_hip.py": ["--diam", "8", "--steps", "300"],
    "turbulent_channel_3d_hip.py": ["--h", "16", "--steps", "300"],
    "windtunnel_3d_hip.py": ["--nx", "96", "--steps", "200", "--every", "100", "--out", "{tmp}"],
    "windtunnel_3d_hip.py --hybrid": ["--nx", "96", "--steps", "200", "--every", "100", "--hybrid"],
    "mlups_3d_hip.py": ["128", "40", "hip", "fp32/fp32", "--repetitions", "2", "--export_final_velocity"],
    "mlups_3d_hip.py --velocity_set D3Q27 --collision_model KBC": ["96", "20", "hip", "fp64/fp32"],
}


@pytest.mark.parametrize("case", list(CASES))
def test_example_runs(case, tmp_path):
    script = case.split()[0]
    extra = case.split()[1:]
    args = [a.replace("{tmp}", str(tmp_path)) for a in CASES[case]] + extra
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + args, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "MLUPS" in res.stdout
    if "{tmp}" in " ".join(CASES[case]):
        files = sorted(os.listdir(tmp_path))
        assert any(f.endswith(".vtk") for f in files) and any(f.endswith(".png") for f in files), files
