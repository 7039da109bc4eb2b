"""CPU-side checks of the drop-in boundary: the C-ABI library loads here (no GPU) and
exports exactly the symbols ``include/rankaae_hip.h`` declares; no compute calls."""
import ctypes
import os
import re

from rankaae_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for fn in os.listdir(os.path.join(REPO, "include")):
        if fn.endswith(".h"):
            with open(os.path.join(REPO, "include", fn)) as f:
                txt = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
            names |= set(re.findall(r"\b(raae_[a-z0-9_]+)\s*\(", txt))
    return names


def test_header_and_binding_agree():
    assert _declared() == set(_lib.SIGNATURES), _declared() ^ set(_lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run rankaae_amd/csrc/build.sh (or __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert _lib.load().raae_abi_version() == _lib.ABI_VERSION
    assert _lib.load().raae_source_digest().decode() == _lib.source_digest(), "stale build"
    # host-side validation rejects bad shapes without touching a GPU
    assert _lib.load().raae_mse_fwd_bwd(None, None, 0, None, None, None, None, None) == -1
    assert b"invalid argument" in _lib.load().raae_error_string(-1)


def test_bn_struct_layout_matches_header():
    # raae_bn_t: ptr, int, float, ptr, ptr, float, float, int  (natural alignment, 48 bytes)
    assert ctypes.sizeof(_lib.BnT) == 48
    assert _lib.BnT.running_mean.offset == 16 and _lib.BnT.momentum.offset == 32


def test_graft_entry_build_runs():
    """The driver's "does it build" hook: compiles (a no-op when the objects are current), loads the library
    and imports the package and the checker."""
    import __graft_entry__ as g
    g.build()
