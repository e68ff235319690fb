"""The C-ABI library loads and exports every symbol include/nmpc.h declares (no compute calls:
this runs without a GPU).  Also: the product path has no CPU fallback."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "nmpc.h")
LIB = os.path.join(ROOT, "mpc_local_planner_amd", "libnmpc_hip.so")


def _declared():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nmpc_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        from mpc_local_planner_amd import build
        build.build_hip()
    return LIB


def test_every_declared_symbol_is_exported(built_lib):
    names = _declared()
    assert len(names) >= 18
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    lib = ctypes.CDLL(built_lib)
    for n in names:
        getattr(lib, n)


def test_binding_lists_the_same_symbols(built_lib):
    from mpc_local_planner_amd import capi
    assert sorted(capi.EXPORTS) == _declared()


def test_config_struct_matches_header_layout(built_lib):
    from mpc_local_planner_amd import capi
    cfg = capi.default_config(40, 0.35)      # pure host call, no GPU needed
    assert (cfg.N, cfg.dt, cfg.wheelbase) == (40, 0.35, 2.65)
    assert list(cfg.W) == [5, 5, 10, 0.01, 0.1, 0.5, 0.1]
    assert cfg.ipm_tol_comp == 1e-11 and cfg.ipm_max_iter == 50


def test_shard_range_matches_the_python_rule(built_lib):
    """nmpc_shard_range (host-only) cuts a fleet exactly as sharding.shard_range does; blocks tile the fleet."""
    import ctypes as C
    from mpc_local_planner_amd import capi, sharding
    l = capi.lib()
    for total in (0, 1, 7, 4096, 1_000_003, 2**33 + 5):
        for world in (1, 2, 3, 8):
            end = 0
            for rank in range(world):
                b, c = C.c_int64(), C.c_int64()
                assert l.nmpc_shard_range(total, rank, world, C.byref(b), C.byref(c)) == capi.OK
                assert (b.value, b.value + c.value) == sharding.shard_range(rank, world, total)
                assert b.value == end
                end = b.value + c.value
            assert end == total
    b, c = C.c_int64(), C.c_int64()
    assert l.nmpc_shard_range(10, 2, 2, C.byref(b), C.byref(c)) == capi.EINVAL
    assert l.nmpc_shard_range(-1, 0, 2, C.byref(b), C.byref(c)) == capi.EINVAL


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under mpc_local_planner_amd/ may import, include
    or link it (and there is no CPU fallback: capi.lib() raises when the HIP library is missing)."""
    pkg = os.path.join(ROOT, "mpc_local_planner_amd")
    pat = re.compile(r"(from\s+oracle|import\s+oracle|oracle/|nmpc_oracle|orc_[a-z_]+\()")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not pat.search(txt), (dp, f, pat.search(txt).group(0))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mpc_local_planner_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "libnmpc_hip.so"))
    with pytest.raises(capi.NmpcError):
        capi.lib()
