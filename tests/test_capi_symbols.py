"""The C-ABI library loads (no GPU needed for that) and exports every symbol the header declares."""
import ctypes
import os
import re

from fluidsolver_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fluidsolver_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fsa_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_host_only_entry_points():
    lib = capi.load()
    assert b"gfx950" in lib.fsa_version()
    p = capi.make_params(capi.PRECISION_F32, spacing=0.005, smoothratio=2.0, vmin=(-0.4, -0.1, -0.4),
                         vmax=(0.4, 1.0, 0.4))
    assert list(p.gridres) == [80, 110, 80]
    assert p.num_cells == 704000
    # -45 / (pi * h^6) and 1 / (pi * (h/2)^3) for h = 0.01, in the reference's mixed precision
    assert abs(p.kspikydiff / (-45.0 / (3.141592 * 1e-12)) - 1) < 1e-5
    assert abs(p.kernel_cubic / (1 / 3.141593 / 1.25e-7) - 1) < 1e-5
    rc = lib.fsa_derive_params(None, 0, 0.005, 2.0, None, None)
    assert rc == -1 and b"NULL" in lib.fsa_last_error()


def test_params_match_oracle_derivation(orc32):
    for vmin, vmax in (((-0.4, -0.1, -0.4), (0.4, 1.0, 0.4)), ((-0.02, -0.02, -0.02), (5.18, 1.98, 1.34))):
        p = capi.make_params(capi.PRECISION_F32, spacing=0.005, smoothratio=2.0, vmin=vmin, vmax=vmax)
        o = orc32.make_params(0.005, 2.0, vmin, vmax)
        assert list(p.gridres) == list(o.gridres)
        for f in ("dx", "smoothradius", "kspikydiff", "kernel_cubic", "kernel_cubic_gradient"):
            assert float(getattr(p, f)) == float(getattr(o, f)), f
