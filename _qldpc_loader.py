"""Registers the hyphen-named package directory `qcrypto-ldpc_amd/` as module `qcrypto_ldpc_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "qcrypto-ldpc_amd")


def load():
    if "qcrypto_ldpc_amd" in sys.modules:
        return sys.modules["qcrypto_ldpc_amd"]
    spec = importlib.util.spec_from_file_location("qcrypto_ldpc_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["qcrypto_ldpc_amd"] = mod
    try:
        spec.loader.exec_module(mod)
    except Exception:
        del sys.modules["qcrypto_ldpc_amd"]
        raise
    return mod
