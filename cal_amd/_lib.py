"""ctypes binding of libcalhip.so, driven by include/cal_hip.h.

The header is the single source of truth: every ``CAL_API`` declaration is
parsed into a ctypes prototype.  There is **no fallback**: if the shared
library is missing ``lib()`` raises.

``libcalhost.so`` (cal_amd/csrc_host/calhost.cpp) implements the operator-level
subset of the same symbols for HOST pointers in plain C++ (the
reference's CPU plumbing run, BASELINE.json configs[0]).  ``call(..., host=True)``
routes there; ``cal_amd.ops`` picks it for CPU-resident tensors only -- data on
the GPU never goes to the host library, and a missing ``libcalhip.so`` stays an
error rather than a reason to compute on the CPU.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "cal_hip.h")
LIB_PATH = os.path.join(HERE, "lib", "libcalhip.so")
HOST_LIB_PATH = os.path.join(HERE, "lib", "libcalhost.so")
#: operator-level entry points libcalhost.so implements (everything a CPU-resident model needs; the step engine,
#: on-device collate and the device permutation draw are GPU-only)
HOST_SYMBOLS = ("cal_last_error", "cal_version", "cal_plan_build", "cal_graph_ptr", "cal_gcn_norm_fwd", "cal_spmm_fwd",
                "cal_colsum_parts", "cal_relu_bwd_colsum", "cal_gcn_norm_bwd", "cal_gemm_ws", "cal_gemm", "cal_gemm_ks",
                "cal_edge_att_fwd", "cal_edge_att_bwd_ws", "cal_edge_att_bwd", "cal_node_att_split_fwd",
                "cal_node_att_bwd_ws", "cal_node_att_split_bwd", "cal_add_pool_fwd", "cal_add_pool_bwd", "cal_gat_fwd",
                "cal_gat_bwd_ws", "cal_gat_bwd", "cal_gat_dropout_mask", "cal_collate_host")

_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64, "float": ctypes.c_float, "double": ctypes.c_double,
    "void": None,
}


def _ctype(decl: str):
    decl = decl.replace("const", " ").strip()
    if "*" in decl:
        return ctypes.c_char_p if decl.replace(" ", "").startswith("char*") else ctypes.c_void_p
    return _SCALARS[decl.split()[0]]


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object], List[str]]]:
    """{name: (restype, [argtypes], [argnames])} for every CAL_API declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"CAL_API\s+([\w\s\*]+?)\s*\b(cal_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                argtypes.append(_ctype(mm.group(1)))
                argnames.append(mm.group(2))
        protos[name] = (_ctype(ret), argtypes, argnames)
    return protos


class CalError(RuntimeError):
    pass


_LIB = None
_PROTOS = None


def lib():
    """Load libcalhip.so (building nothing: use ``python -m cal_amd.build``)."""
    global _LIB, _PROTOS
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise CalError(
            "libcalhip.so not found at %s -- build it with `python -m cal_amd.build` "
            "(there is no CPU fallback)" % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    _PROTOS = parse_header()
    for name, (ret, argtypes, _) in _PROTOS.items():
        fn = getattr(handle, name)      # AttributeError if the library lacks a declared symbol
        fn.restype = ret
        fn.argtypes = argtypes
    _LIB = handle
    return _LIB


def protos():
    lib()
    return _PROTOS


_HOST = None


def host_lib():
    """Load libcalhost.so (the host implementation of HOST_SYMBOLS)."""
    global _HOST
    if _HOST is not None:
        return _HOST
    if not os.path.exists(HOST_LIB_PATH):
        raise CalError("libcalhost.so not found at %s -- build it with `python -m cal_amd.build`" % HOST_LIB_PATH)
    handle = ctypes.CDLL(HOST_LIB_PATH)
    pr = parse_header()
    for name in HOST_SYMBOLS:
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = pr[name][0], pr[name][1]
    _HOST = handle
    return _HOST


def call(name: str, *args, host: bool = False):
    """Call an int-returning entry point; raise CalError with cal_last_error() on failure.
    ``host``: the pointers are HOST pointers -> libcalhost.so (operator-level entry points only)."""
    if host and name not in HOST_SYMBOLS:
        raise CalError("%s has no host implementation (GPU only)" % name)
    h = host_lib() if host else lib()
    rc = getattr(h, name)(*args)
    if rc != 0:
        raise CalError("%s failed (%d): %s" % (name, rc, h.cal_last_error().decode()))


def query(name: str, *args, host: bool = False) -> int:
    """Call a size-query entry point (returns int64)."""
    return int(getattr(host_lib() if host else lib(), name)(*args))
