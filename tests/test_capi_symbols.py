"""The C-ABI library loads and exports every symbol include/ipc_amd.h declares (no compute
calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ipc_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ipc_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from ipc_amd import capi
    lib = capi.load()
    decl = _declared()
    assert len(decl) >= 15
    for name in decl:
        assert hasattr(lib, name), name
    assert sorted(capi.SYMBOLS) == decl
    assert ctypes.sizeof(capi.CellInfo) == 48


def test_argument_errors_are_reported_not_thrown():
    from ipc_amd import capi
    lib = capi.load()
    assert lib.ipc_rows_per_rank(10, 4) == 3
    assert lib.ipc_rows_per_rank(1256, 8) == 157
    # NULL handle => negative status + message, no crash, no GPU needed
    assert lib.ipc_set_candidates(None, 0, None, None, None) == -1
    assert b"NULL handle" in lib.ipc_last_error()
    assert lib.ipc_solve_rows(None, 0, 1, None, None) == -1
    assert lib.ipc_destroy(None) == 0


def test_product_package_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under ipc_amd/ may import, link or call it."""
    pkg = os.path.join(ROOT, "ipc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "ipc_oracle" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
