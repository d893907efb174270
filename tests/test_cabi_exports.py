"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly the
entry points include/egom2p_hip.h declares (no compute is launched without a GPU)."""
import os
import re
import subprocess

from egom2p_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "egom2p_hip.h")).read()
    return set(re.findall(r"^\s*int\s+(ego_\w+)\s*\(", text, flags=re.M))


def test_header_and_binding_agree():
    assert _declared() == set(L.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = L.load()
    for name in _declared():
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ego_\w+)", out))
    assert _declared() <= exported
    assert lib.ego_abi_version() == 1


def test_no_oracle_import_in_product_path():
    """The product package must never import the oracle (a fallback would void parity claims)."""
    pkg = os.path.join(ROOT, "egom2p_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
