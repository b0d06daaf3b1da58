"""The C-ABI library builds, loads, exports every symbol include/i2t.h declares, and the ctypes binding matches
the header argument for argument.  No compute call is made (runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from image2text_amd import lib as i2tlib
from image2text_amd.build import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'i2t.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\bint\s+(i2t_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        args = [a.strip() for a in m.group(2).replace('\n', ' ').split(',')]
        out[m.group(1)] = [] if args == ['void'] else args
    return out


def ctype_of(decl: str):
    decl = decl.strip()
    if decl.endswith('*') or '*' in decl:
        if decl.startswith('char*') or decl.startswith('char *'):
            return C.c_char_p
        if 'void**' in decl.replace(' ', ''):
            return C.POINTER(C.c_void_p)
        return C.c_void_p
    base = decl.rsplit(' ', 1)[0].strip()
    return {'int': C.c_int, 'long': C.c_long, 'float': C.c_float, 'int64_t': C.c_int64, 'size_t': C.c_size_t,
            'unsigned': C.c_uint}[base]


@pytest.fixture(scope='module')
def built():
    return build_library()


def test_library_builds_and_loads(built):
    assert os.path.exists(built)
    lib = i2tlib.load()
    assert lib.i2t_abi_version() == i2tlib.ABI_VERSION


def test_every_declared_symbol_is_exported_and_bound(built):
    decl = header_functions()
    assert len(decl) >= 30
    lib = C.CDLL(built)
    for name in decl:
        assert hasattr(lib, name), f'{name} declared in i2t.h but not exported'
    assert set(decl) == set(i2tlib.SIGNATURES), set(decl) ^ set(i2tlib.SIGNATURES)


def test_binding_matches_header_argument_types(built):
    decl = header_functions()
    for name, args in decl.items():
        want = [ctype_of(a) for a in args]
        got = i2tlib.SIGNATURES[name]
        assert len(want) == len(got), f'{name}: header has {len(want)} args, binding {len(got)}'
        for i, (w, g) in enumerate(zip(want, got)):
            assert w is g or (w is C.c_void_p and g is C.c_void_p), f'{name} arg {i} ({args[i]}): {w} vs {g}'


def test_error_channel(built):
    lib = i2tlib.load()
    # argument validation happens before any HIP call: a null operand must be refused with a message
    rc = lib.i2t_gemm_bf16(None, None, 8, 0, None, 8, 0, None, 8, 0, 1, 1, 8, 1.0, None, 0, None, 0, None, 0, None, 0, 0, 0, 0, 0, 1.0)
    assert rc == -1
    assert 'null operand' in i2tlib.last_error()
