"""ctypes loader for libmusicxl.so (the C-ABI boundary, include/musicxl.h).

There is NO fallback: if the library is missing or a call returns non-zero, we raise.  A product path that silently
ran on PyTorch/CPU would void every parity claim.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MXL_LIB_PATH') or os.path.join(_HERE, 'libmusicxl.so')   # override: A/B builds of the library
HEADER_PATH = os.path.join(_HERE, '..', 'include', 'musicxl.h')

_lib = None


class MusicXLError(RuntimeError):
    pass


_CTYPE = {
    'int': C.c_int, 'unsigned': C.c_uint, 'float': C.c_float, 'unsigned long long': C.c_ulonglong,
    'long long': C.c_longlong, 'size_t': C.c_size_t,
}


def declared_functions(header_path: str = HEADER_PATH):
    """Parse `include/musicxl.h` -> {name: (restype, [argtypes])}.  The header is the single source of truth."""
    src = open(header_path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'(const char\*|int|size_t)\s+(mxl_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                if '*' in a:
                    argtypes.append(C.c_void_p)
                else:
                    ty = ' '.join(a.split(' ')[:-1])
                    ty = ty.replace('const ', '')
                    argtypes.append(_CTYPE[ty])
        restype = C.c_char_p if ret.startswith('const char') else (C.c_size_t if ret == 'size_t' else C.c_int)
        out[name] = (restype, argtypes)
    return out


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MusicXLError(
                f'{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc, gfx950). '
                'There is no CPU / PyTorch fallback for the product path.')
        # torch first: it ships its own libamdhip64, and the library must bind to THAT copy of the HIP runtime -- loaded before torch it
        # pulls in /opt/rocm's, the process then holds two runtimes, and every call through the second one fails with
        # hipErrorNoDevice (seen with __graft_entry__.build() followed by smoke() in one process)
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in declared_functions().items():
            fn = getattr(_lib, name)  # AttributeError if the header declares something the .so lacks
            fn.restype = restype
            fn.argtypes = argtypes
        if _lib.mxl_abi_version() != 1:
            raise MusicXLError('libmusicxl ABI version mismatch')
    return _lib


def check(code: int, what: str = ''):
    if code != 0:
        msg = lib().mxl_error_string(code)
        raise MusicXLError(f'{what} failed with code {code}: {msg.decode() if msg else "?"}')
