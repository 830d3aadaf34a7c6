"""CPU-side checks of the C-ABI boundary: the built library loads and exports every symbol include/csmae.h declares
(no kernel is launched here), the Python binding table matches the header, and the product refuses to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "csmae.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:const char\*|int)\s+(csmae_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        out[m.group(1)] = [] if args == ["void"] else args
    return out


def test_library_exports_every_declared_symbol():
    import csmae_hip
    assert os.path.exists(csmae_hip.LIB_PATH), "run `make` / __graft_entry__.build() first"
    lib = ctypes.CDLL(csmae_hip.LIB_PATH)
    decl = header_functions()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/csmae.h but not exported"
    assert set(decl) == set(csmae_hip.exported_symbols())
    # ... and nothing else: the dynamic symbol table of the library (`nm -D`) holds exactly the declared csmae_* functions
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", csmae_hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split()[-2:-1] == ["T"] and ln.split()[-1].startswith("csmae_")}
    assert exported == set(decl), (sorted(exported - set(decl)), sorted(set(decl) - exported))
    lib.csmae_abi_version.restype = ctypes.c_int
    assert lib.csmae_abi_version() == csmae_hip.ABI_VERSION == 7


def test_binding_arity_matches_header():
    import csmae_hip
    decl = header_functions()
    for name, sig in csmae_hip._SIGNATURES.items():
        assert len(sig) == len(decl[name]), (name, len(sig), decl[name])
        for ct, arg in zip(sig, decl[name]):
            if "*" in arg:
                assert ct is ctypes.c_void_p, (name, arg)
            elif arg.startswith("long long"):
                assert ct is ctypes.c_longlong, (name, arg)
            elif arg.startswith("float"):
                assert ct is ctypes.c_float, (name, arg)
            else:
                assert ct is ctypes.c_int, (name, arg)


def test_ops_refuse_cpu_tensors():
    from csmae_hip import ops
    a = torch.zeros(8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(a, a, torch.zeros(8, 8))
