"""The C-ABI library: loads without a GPU, exports every symbol the public
header declares, argument validation returns error codes (no compute calls
here), and its host-side mu-law tables equal the oracle's."""
import ctypes
import os
import re

import numpy as np
import pytest

from util import O, ROOT

HEADER = os.path.join(ROOT, 'include', 'wavenet_hip.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(wn_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported(hip_lib):
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(hip_lib, s), 'missing export %s' % s


def test_binding_table_matches_header(hip_lib):
    from wavenet import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_error_strings(hip_lib):
    assert hip_lib.wn_version() >= 100
    for code in (0, -1, -2, -3, -4, -5):
        assert len(hip_lib.wn_error_string(code)) > 0


def test_argument_validation_without_gpu(hip_lib):
    # all of these return before any launch
    assert hip_lib.wn_layer_fwd(None, None, None, None, None, None, None, 0,
                                1, 8, 1, 1, 1, None) == -5       # NULL
    assert hip_lib.wn_causal_gather(None, None, None, 1, 1, 1, 2, 32, None) == -5
    buf = (ctypes.c_float * 64)()
    a = ctypes.addressof(buf)
    assert hip_lib.wn_gemm_nn(a, 8, 0, 0, a, 8, None, None, 0, None, 0, a, 8,
                              0, 0, None, 0, 8, 8, 0, None) == -1  # M <= 0
    assert hip_lib.wn_gemm_nn(a, 8, 0, 0, a, 8, None, None, 0, None, 0, a, 8,
                              0, 0, None, 4, 6, 8, 0, None) == -2  # N % 4
    assert hip_lib.wn_gemm_nn(a + 4, 8, 0, 0, a, 8, None, None, 0, None, 0, a,
                              8, 0, 0, None, 4, 8, 8, 0, None) == -3  # align
    assert hip_lib.wn_xent(a, 6, a, a, a, 1, 1, 6, 1, None) == -2
    assert hip_lib.wn_layer_bwd2(None, None, None, None, None, None, None, None,
                                 None, None, 0, 8, 1, None) == -1
    assert hip_lib.wn_mu_law_thresholds_host(1, a) == -1
    assert hip_lib.wn_fastgen_run(a, a, 0, a, None, a, None, a, None, None, a,
                                  2, 1024, 16, a, a, a, 1, 1, 1.0, 0, None, 1,
                                  0, 1, None) == -2              # S > 512


def test_stack_entries_validate_without_gpu(hip_lib):
    """wn_stack_fwd / wn_stack_bwd: sizes and argument checks (no launch)."""
    # (one flag per 16-row tile: enough for either tile height)
    assert hip_lib.wn_stack_flag_count(8, 16000, 50) == 50 * 8 * 1000
    assert hip_lib.wn_stack_flag_count(2, 33, 3) == 3 * 2 * 3     # ragged tile counts
    assert hip_lib.wn_stack_flag_count(0, 33, 3) == 0
    # tile height: 16 rows while the batch has at most four 32-row tiles per CU
    # (256 CUs assumed without a device)
    assert hip_lib.wn_stack_tile_rows(8, 16000, 0) == 32
    assert hip_lib.wn_stack_tile_rows(3, 16000, 0) == 32
    assert hip_lib.wn_stack_tile_rows(2, 16000, 0) == 16
    assert hip_lib.wn_stack_tile_rows(1, 16000, 0) == 16
    assert hip_lib.wn_stack_tile_rows(0, 16000, 0) == 32
    assert hip_lib.wn_stack_bwd_slabs(8, 16000, 0) == 250           # 16 tiles per group
    assert hip_lib.wn_stack_bwd_slabs(1, 16000, 0) == 125           # 8 waves x one 16-row tile
    assert hip_lib.wn_stack_bwd_slabs(0, 16000, 0) == 0
    # the explicit variant word (WN_STACK_VARIANT of wavenet_hip.h): tile rows
    # in bits 0..5, waves per workgroup in bits 8..11 -- the only thing besides
    # the shape that selects a launch (the library reads no environment)
    assert hip_lib.wn_stack_tile_rows(8, 16000, 16) == 16
    assert hip_lib.wn_stack_tile_rows(1, 16000, 32) == 32
    assert hip_lib.wn_stack_tile_rows(1, 16000, 17) == 16         # not a height: the shape's
    assert hip_lib.wn_stack_bwd_slabs(1, 16000, 16 | (4 << 8)) == 250   # 4 waves x one 16-row tile
    assert hip_lib.wn_stack_bwd_slabs(1, 16000, 32) == 125       # 32 rows: 4 waves x one tile
    assert hip_lib.wn_stack_bwd_slabs(1, 16000, 32 | (8 << 8)) == 63    # ... 8 waves forced
    buf = (ctypes.c_float * 64)()
    a = ctypes.addressof(buf)
    assert hip_lib.wn_stack_fwd(None, a, a, a, None, 0, 0, a, a, a, None,
                                2, 1, 64, 1, 0, None) == -5
    assert hip_lib.wn_stack_fwd(a, a, None, a, None, 0, 0, a, a, a, None,
                                2, 1, 64, 1, 0, None) == -5          # save_sg without SG
    assert hip_lib.wn_stack_fwd(a, a, a, a, None, 0, 0, a, a, a, None,
                                0, 1, 64, 1, 0, None) == -1          # L <= 0
    assert hip_lib.wn_stack_pack(a, 100, a, a, 2, None) == -1     # stride < block
    assert hip_lib.wn_stack_pack(a, 5216, None, None, 2, None) == -5
    assert hip_lib.wn_stack_wimg_floats() % 256 == 0
    assert hip_lib.wn_stack_fwd(a + 4, a, a, a, None, 0, 0, a, a, a, None,
                                2, 1, 64, 1, 0, None) == -3
    assert hip_lib.wn_stack_fwd(a, a, a, a, None, 0, 0, a, a, a, None,
                                257, 1, 64, 1, 0, None) == -2        # L > 256
    pl = 1 * 64 * 32
    assert hip_lib.wn_stack_bwd(a, a, a, a, None, pl, a, a, a, 5216, None, a, a, a,
                                None, 2, 1, 64, 0, None) == -5
    assert hip_lib.wn_stack_bwd(a, a, a, a, a, pl, a, a, a, 5216, None, a, a, a,
                                None, 2, 0, 64, 0, None) == -1
    assert hip_lib.wn_stack_bwd(a, a, a, a, a, pl, a, a, a, 100, None, a, a, a,
                                None, 2, 1, 64, 0, None) == -1       # slab stride too small
    assert hip_lib.wn_stack_bwd(a, a + 4, a, a, a, pl, a, a, a, 5216, None, a, a, a,
                                None, 2, 1, 64, 0, None) == -3
    assert hip_lib.wn_stack_bwd(a, a, a, a, a, pl, a + 4, a, a, 5216, None, a, a, a,
                                None, 2, 1, 64, 0, None) == -3       # Q misaligned
    # the q planes are not optional
    assert hip_lib.wn_stack_bwd(a, a, a, a, a, 0, None, a, a, 5216, None, a, a, a,
                                None, 2, 1, 64, 0, None) == -5
    assert hip_lib.wn_stack_bwd(a, a, a, a, a, 77, a, a, a, 5216, None, a, a, a,
                                None, 2, 1, 64, 0, None) == -1       # neither 0 nor a plane


@pytest.mark.parametrize('q', [2, 16, 123, 128, 256])
def test_host_tables_equal_oracle(hip_lib, q):
    thr = np.empty(q - 1, np.float32)
    lut = np.empty(q, np.float32)
    assert hip_lib.wn_mu_law_thresholds_host(q, thr.ctypes.data) == 0
    assert hip_lib.wn_mu_law_decode_table_host(q, lut.ctypes.data) == 0
    assert np.array_equal(thr, O.mu_law_thresholds(q))
    assert np.array_equal(lut, O.mu_law_decode(np.arange(q), q))


def test_product_does_not_import_oracle():
    """The shipped package must never route through the oracle."""
    pkg = os.path.join(ROOT, 'tensorflow-wavenet_amd', 'wavenet')
    for f in os.listdir(pkg):
        if f.endswith('.py'):
            src = open(os.path.join(pkg, f)).read()
            assert 'oracle' not in src, f


def test_library_reads_no_environment(hip_lib):
    """Kernel selection is a function of the arguments (the `variant` word of
    the stack launches, include/wavenet_hip.h) -- the shared object does not
    even import getenv, and no source mentions it."""
    import subprocess
    from wavenet import _lib
    out = subprocess.run(['nm', '-D', '--undefined-only', _lib.LIB_PATH],
                         capture_output=True, text=True, check=True).stdout
    assert 'getenv' not in out
    csrc = os.path.join(ROOT, 'tensorflow-wavenet_amd', 'csrc')
    for f in os.listdir(csrc):
        assert 'getenv' not in open(os.path.join(csrc, f)).read(), f


def test_missing_library_fails_loudly(monkeypatch):
    from wavenet import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libwavenet_hip.so')
    with pytest.raises(_lib.WaveNetHipError):
        _lib.load()


def test_persistent_generator_role_mapping_is_a_permutation(hip_lib):
    """fg_persist_kernel's workgroup -> role mapping (host-side mirror
    wn_fastgen_persist_role of the kernel's own function): a permutation of the
    roles for every grid, the chain's roles -- segments 0 .. nseg - 1 and the
    draw (= total - 1) -- on the blocks 0, 8, 16, ... (one XCD) when the grid is
    large enough, the identity otherwise."""
    for total, nseg in ((86, 5), (41, 5), (42, 5), (40, 5), (27, 2), (5, 1), (9, 1),
                        (10, 1), (100, 8), (65, 8), (66, 8), (2, 1)):
        roles = [hip_lib.wn_fastgen_persist_role(b, total, nseg) for b in range(total)]
        assert sorted(roles) == list(range(total)), (total, nseg)
        if total > 8 * nseg:
            for k in range(nseg):
                assert roles[8 * k] == k
            assert roles[8 * nseg] == total - 1
            # the other roles keep their order
            rest = [r for b, r in enumerate(roles) if not (b % 8 == 0 and b // 8 <= nseg)]
            assert rest == sorted(rest)
        else:
            assert roles == list(range(total))
    assert hip_lib.wn_fastgen_persist_role(5, 5, 1) == -1       # block out of range
    assert hip_lib.wn_fastgen_persist_role(0, 1, 1) == -1       # no room for chain + draw
