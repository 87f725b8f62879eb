"""CPU: the C-ABI shared library loads and exports every symbol include/aoadmm_hip.h declares;
struct layouts agree with the ctypes mirror; compute calls fail loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'aoadmm_hip.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(aoadmm_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_list_agree(pkg):
    assert declared_functions() == sorted(pkg.SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    if not os.path.exists(pkg.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = pkg.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.aoadmm_abi_version() == 3


def test_struct_layouts(pkg):
    capi = __import__('importlib').import_module('matlab-code_amd._capi')
    # aoadmm_options: 2*int32, 6*double, int32(+pad), double, 2*int32, double, int32, 7*int32
    assert C.sizeof(capi.Options) == 8 + 48 + 8 + 8 + 8 + 8 + 4 + 28 + 0 or C.sizeof(capi.Options) % 8 == 0
    assert capi.Options.AbsFuncTol.offset == 8 and capi.Options.bsum.offset == 56
    assert capi.Options.bsum_weight.offset == 64 and capi.Options.increase_factor_rhoBk.offset == 80
    assert capi.Result.func_val_conv.offset == 56


def test_constraint_ids_match_header(pkg):
    text = open(HEADER).read()
    ids = {m.group(1): int(m.group(2)) for m in re.finditer(r'AOADMM_C_([A-Z0-9_]+)\s*=\s*(\d+)', text)}
    assert ids['NONNEG'] == pkg.CONSTRAINT_IDS['non-negativity']
    assert ids['TV'] == pkg.CONSTRAINT_IDS['TV regularization']
    assert ids['NONNEG_L2_SPHERE'] == pkg.CONSTRAINT_IDS['non-negative l2-sphere']
    assert ids['TPARAFAC2'] == pkg.CONSTRAINT_IDS['tPARAFAC2']
    assert len(pkg.CONSTRAINT_IDS) == 20


def test_no_cpu_fallback(pkg):
    """Without a GPU the product path must fail loudly (never route through the oracle)."""
    lib = pkg.load_library()
    n = C.c_int(-1)
    assert lib.aoadmm_device_count(C.byref(n)) == 0
    if n.value == 0:
        with pytest.raises(pkg.AoadmmError) as ei:
            pkg.Engine(0)
        assert 'no CPU fallback' in str(ei.value)


def test_product_package_never_imports_oracle():
    pkgdir = os.path.join(ROOT, 'matlab-code_amd')
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp', '.m')):
                src = open(os.path.join(dirpath, f), errors='replace').read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_host_layer_rejects_what_the_device_path_does_not_cover(pkg):
    with pytest.raises(pkg.UnsupportedOnDevice):
        pkg.constraint_descriptor(('custom', lambda x, rho: x))
    with pytest.raises(ValueError):
        pkg.constraint_descriptor(('no such constraint',))
    assert pkg.row_block(2000, 8, 7) == (1750, 250) and pkg.row_block(10, 4, 3) == (9, 1)


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """The boundary is a C ABI: include/aoadmm_hip.h must compile as C99 (no C++ leaks) and a plain-C caller
    (examples/solve_cp.c, the call sequence of a MEX gateway) must link against the library.  Compute needs a GPU."""
    import subprocess
    src = tmp_path / 'hdr.c'
    src.write_text('#include "aoadmm_hip.h"\nint main(void) { return aoadmm_abi_version() == 0; }\n')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only',
                    '-I', os.path.join(ROOT, 'include'), str(src)], check=True)
    exe = tmp_path / 'solve_cp'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-O1', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'examples', 'solve_cp.c'), '-L', os.path.join(ROOT, 'matlab-code_amd'),
                    '-laoadmm_hip', '-lm', '-o', str(exe)], check=True)
    assert exe.exists()
