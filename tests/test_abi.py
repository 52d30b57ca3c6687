"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares; the ctypes binding covers exactly that set.  No compute call is made (no GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

from lambdapic_amd import _lib

ROOT = Path(__file__).resolve().parents[1]


def header_symbols():
    text = (ROOT / "include" / "lambdapic_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lpa_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_symbols():
    syms = header_symbols()
    assert "lpa_push_deposit_tiled_2d" in syms and "lpa_fdtd_e_2d" in syms
    assert len(syms) >= 25


def test_library_exports_every_declared_symbol():
    assert _lib.LIB_PATH.exists(), "run __graft_entry__.build() / python -m lambdapic_amd.build"
    L = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [s for s in header_symbols() if not hasattr(L, s)]
    assert not missing, missing


def test_binding_matches_header():
    assert sorted(_lib.SIGNATURES) == header_symbols()
    L = _lib.lib()
    assert L.lpa_version() >= 100
    assert L.lpa_last_error() is not None


def test_struct_layout_matches_header():
    # field counts / sizes of the POD structs (x86-64 SysV, natural alignment)
    assert ctypes.sizeof(_lib.lpa_grid) == 4 * 4 + 6 * 8 + 10 * 8
    assert ctypes.sizeof(_lib.lpa_particles) == 8 + 8 * 8 + 6 * 8 + 8 + 8
    assert ctypes.sizeof(_lib.lpa_push_params) == 3 * 8 + 8 + 12 * 8 + 3 * 8 + 5 * 8 + 8      # wrap + flags share 8 bytes
    # 2 i32, i64, 2 i32, 5 pointers, 2 i32, 8 pointers, 5 pointers, 2 i32
    assert ctypes.sizeof(_lib.lpa_tiling) == 8 + 8 + 8 + 5 * 8 + 8 + 8 * 8 + 5 * 8 + 8


def test_struct_layout_matches_the_c_compiler(tmp_path):
    """sizeof / offsetof of every struct of the header as gcc lays them out == the ctypes mirror"""
    import subprocess
    structs = {"lpa_grid": _lib.lpa_grid, "lpa_particles": _lib.lpa_particles, "lpa_tiling": _lib.lpa_tiling,
               "lpa_push_params": _lib.lpa_push_params, "lpa_cpml_axis": _lib.lpa_cpml_axis,
               "lpa_free_slots": _lib.lpa_free_slots, "lpa_step_species": _lib.lpa_step_species,
               "lpa_step_desc": _lib.lpa_step_desc, "lpa_face_msg": _lib.lpa_face_msg,
               "lpa_step_migrate": _lib.lpa_step_migrate, "lpa_step_slab": _lib.lpa_step_slab}
    lines = []
    for name, cls in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f in cls._fields_:
            lines.append(f'printf("{name}.{f[0]} %zu\\n", offsetof({name}, {f[0]}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lambdapic_amd.h"\nint main(void) {\n'
                   + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(out[name]) == ctypes.sizeof(cls), name
        for f in cls._fields_:
            assert int(out[f"{name}.{f[0]}"]) == getattr(cls, f[0]).offset, (name, f[0])


def test_bad_arguments_are_reported_not_crashed():
    """argument errors come back as a status + message, like the reference's PyArg_ParseTuple
    failures (no other error reporting exists there)"""
    L = _lib.lib()
    g = _lib.lpa_grid()   # all zero: invalid
    assert L.lpa_fdtd_b_2d(ctypes.byref(g), 1e-17, None) == -1
    assert b"bad grid" in L.lpa_last_error()
    with pytest.raises(_lib.LpaError):
        _lib.check(-1, "x")


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lambdapic_amd import kernels
    from lambdapic_amd.fields import Fields2D
    with pytest.raises(_lib.LpaError):
        kernels.update_bfield_2d(Fields2D(8, 8, 1e-8, 1e-8, 0, 0, 3), 1e-17)


def test_every_abi_function_is_mapped_in_integration_md():
    """INTEGRATION.md names the reference call each entry point replaces: no entry point may be missing there"""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    names = sorted(set(re.findall(r"\b(lpa_[a-z0-9_]+)\s*\(", (root / "include" / "lambdapic_amd.h").read_text())))
    doc = (root / "INTEGRATION.md").read_text()
    missing = [n for n in names if n not in doc]
    assert not missing, missing
