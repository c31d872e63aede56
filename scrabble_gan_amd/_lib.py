"""ctypes binding of libscrabble_hip.so.  Signatures are parsed from include/scrabble_hip.h so the
Python side can never drift from the declared C-ABI.  There is NO fallback: if the shared library
is missing or a symbol is absent, importing the ops raises."""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "scrabble_hip.h")
LIB_PATH = os.path.join(HERE, "lib", "libscrabble_hip.so")

_CTYPE = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
}


class ScrabbleHipError(RuntimeError):
    pass


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every function the header declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    decls = {}
    for m in re.finditer(r"\b(int|long)\s+(sg_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes, argnames = [], []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
                argnames.append(a.split("*")[-1].strip())
            else:
                parts = a.split()
                argtypes.append(_CTYPE[parts[-2]])
                argnames.append(parts[-1])
        decls[name] = (_CTYPE[ret], argtypes, argnames)
    return decls


_lib = None
_decls = None


def lib():
    global _lib, _decls
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScrabbleHipError(
            "libscrabble_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback." % LIB_PATH)
    l = ctypes.CDLL(LIB_PATH)
    _decls = parse_header()
    for name, (ret, argtypes, _) in _decls.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise ScrabbleHipError("libscrabble_hip.so does not export %s declared in %s" % (name, HEADER)) from e
        fn.restype = ret
        fn.argtypes = argtypes
    _lib = l
    return _lib


_ERR = {-1: "SG_ERR_ARG (bad argument / unsupported shape)", -2: "SG_ERR_LAUNCH (kernel launch failed)",
        -3: "SG_ERR_UNSUPPORTED"}


def call(name: str, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    fn = getattr(lib(), name)
    if len(args) != len(fn.argtypes):           # (cdecl: ctypes itself lets surplus arguments through)
        raise TypeError("%s takes %d arguments (%s), %d given" % (name, len(fn.argtypes), ", ".join(_decls[name][2]), len(args)))
    rc = fn(*args)
    if rc != 0:
        raise ScrabbleHipError("%s failed: %s" % (name, _ERR.get(rc, rc)))
