"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that include/lp_hip.h declares,
and the ctypes signature table agrees with the header (argument count and pointer/int/float kinds).  No GPU calls."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_header():
    src = open(os.path.join(ROOT, 'include', 'lp_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\n\s*((?:const\s+)?[A-Za-z_][\w\s\*]*?)\b(lp_\w+)\s*\(([^;{]*?)\)\s*;', src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ('', 'void') else [a.strip() for a in args.split(',')]
        decls[name] = (ret, params)
    return decls


def kind(ctype_decl):
    if '*' in ctype_decl:
        return 'ptr'
    if re.search(r'\bfloat\b', ctype_decl):
        return 'float'
    if 'long long' in ctype_decl:
        return 'll'
    return 'int'


def test_library_exports_every_declared_symbol():
    from latent_pose_reenactment_amd import _lib
    decls = parse_header()
    assert len(decls) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), f'{name} declared in lp_hip.h but not exported by liblp_hip.so'
    assert set(_lib.SIGNATURES) == set(decls), set(_lib.SIGNATURES) ^ set(decls)


def test_ctypes_signatures_match_header():
    from latent_pose_reenactment_amd import _lib
    cmap = {ctypes.c_void_p: 'ptr', ctypes.c_int: 'int', ctypes.c_float: 'float', ctypes.c_longlong: 'll', ctypes.c_char_p: 'ptr'}
    for name, (ret, params) in parse_header().items():
        res, args = _lib.SIGNATURES[name]
        assert [cmap[a] for a in args] == [kind(p) for p in params], f'{name}: ctypes {args} vs header {params}'
        assert cmap[res] == kind(ret + ' '), (name, ret)


def test_abi_version_and_error_string():
    from latent_pose_reenactment_amd import _lib
    l = _lib.lib()
    assert l.lp_abi_version() == 12
    # argument validation happens before any device work, so it is callable without a GPU
    rc = l.lp_pack_weights(None, None, None, 1, 1, 1, 128, 64, 0, 0, None)
    assert rc == -1 and b'null' in l.lp_last_error()


def test_descriptor_struct_sizes_match_the_python_packers():
    """the device descriptor tables are packed with struct.pack on the host (nn.SNBatch '<QQQQQQQiifi', optim._build_table,
    hipops.PackBatch '<QQQiiiiiiii' / '<QQQQQiiiiiiiiii'): their byte sizes must equal the C structs'"""
    import struct
    from latent_pose_reenactment_amd import _lib
    l = _lib.lib()
    assert l.lp_sn_desc_bytes() == struct.calcsize('<QQQQQQQiifi') == 72
    assert l.lp_pack_desc_bytes() == struct.calcsize('<QQQiiiiiiii') == 56
    assert l.lp_pack_pair_desc_bytes() == struct.calcsize('<QQQQQiiiiiiiiii') == 80
    assert l.lp_mt_desc_bytes() == struct.calcsize('<QQQQq') == 40
    assert l.lp_sn_row_block() == 32 and l.lp_l1_partial_blocks() == 1024
