"""The C++ mirror of the reference's solver classes (include/mpc_local_planner_amd/*.hpp) run on
the GPU through the C ABI and compared with the oracle.  The binary is built by
__graft_entry__.build() (g++, links libnmpc_hip.so)."""
import os
import subprocess

import numpy as np
import pytest

from mpc_local_planner_amd import capi, scenarios as sc
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_mirror")


def test_cpp_mirror_single_and_batch(tmp_path):
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    N, B = 30, 8                                  # the reference's default horizon (mpc_params.yaml:35)
    x0, p = sc.config2(B)
    inp = tmp_path / "in.txt"
    np.savetxt(inp, np.hstack([x0, p]), fmt="%.17g")
    out = subprocess.run([EXE, str(inp), str(N)], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    ocfg = orc.config_like(capi.default_config(N, 0.35))
    xo, uo, so, _ = orc.rti_batch(ocfg, x0, p)
    lines = out.stdout.strip().splitlines()
    assert "ctor-throws ok" in lines[-1]
    nb = 0
    single = []
    for ln in lines:
        t = ln.split()
        if t[0] == "batch":
            b = int(t[1])
            assert t[2] == "1" and float(t[3]) == -1.0          # success, cost = -1 (AcadosSolver.cpp:72)
            h = np.array(t[4:], dtype=float).reshape(N, 8)      # N entries: x_0..x_{N-1}, u_0..u_{N-1}
            s = max(1.0, np.abs(xo[b]).max())
            assert np.abs(h[:, :6] - xo[b, :N]).max() <= 1e-6 * s
            assert np.abs(h[:, 6:] - uo[b]).max() <= 1e-6 * s
            nb += 1
        elif t[0] == "pts":
            # PathTrackingSys::solve(state, pitch, vRef): starts at the car, moves along the track
            assert t[1] == "1"
            pts = np.array(t[3:], dtype=float).reshape(N, 2)
            assert np.abs(pts[0] - [41.0, 0.0]).max() < 1e-9
            assert pts[5, 1] > 3.0 and abs(pts[5, 0] - 40.0) < 3.0
        elif t[0] == "single":
            assert t[2] == "1"
            single.append(np.array(t[4:], dtype=float).reshape(N, 8))
    assert nb == B and len(single) == 3
    # single-instance protocol: cold call, then two warm calls without shift
    x1, u1 = xo[0:1].copy(), uo[0:1].copy()
    for call in range(3):
        if call:
            x1, u1, _, _ = orc.rti_batch(ocfg, x0[0:1], p[0:1], x1, u1, cold=False)
        s = max(1.0, np.abs(x1).max())
        assert np.abs(single[call][:, :6] - x1[0, :N]).max() <= 1e-6 * s, call
        assert np.abs(single[call][:, 6:] - u1[0]).max() <= 1e-6 * s, call
