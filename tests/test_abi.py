"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol."""
import ctypes
import os
import re

from fgt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "fgt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fgt_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert len(syms) >= 20
    assert sorted(_lib.SIGNATURES) == syms


def test_library_exports_every_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python -m fgt_amd.build"
    h = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(h, s), f"{s} not exported"
    assert _lib.lib().fgt_abi_version() == _lib.ABI_VERSION == 9


def test_struct_sizes_match_header():
    # 44 ints/floats + 3 long long + the 10 ints of ABI 7 in fgt_conv_desc (44 * 4 = 176: no padding before the 8-byte fields; 240 bytes: none behind
    # them), 21 ints in fgt_attn_desc
    # ABI 8: + dual_n0, reserved8 (ints; ABI 9: reserved8 became ps_phase_pad) and gb_x0, gb_w, gb_o (long long): 272 bytes, no implicit padding
    assert ctypes.sizeof(_lib.ConvDesc) == 44 * 4 + 3 * 8 + 10 * 4 + 2 * 4 + 3 * 8
    assert _lib.ConvDesc.gb_x0.offset == 248 and _lib.ConvDesc.dual_n0.offset == 240
    assert ctypes.sizeof(_lib.AttnDesc) == 22 * 4 + 8 + 2 * 4 + 5 * 8 + 2 * 4          # ... in_split, tq | ps* | compact, pad_row


def test_rejects_bad_arguments_without_gpu():
    # argument validation happens before any HIP call: callable on a CPU-only box
    d = _lib.ConvDesc()
    rc = _lib.lib().fgt_conv2d(ctypes.byref(d), None, None, None, None, None, None, None, None, None, None)
    assert rc == -1
    assert b"null" in _lib.lib().fgt_last_error()


def test_ops_fail_loudly_on_cpu_tensors():
    import pytest
    import torch
    from fgt_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.layernorm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))


def test_bench_and_smoke_fail_loudly_without_a_gpu():
    """No CPU fallback anywhere on the product path: bench.py refuses to run, smoke() asserts."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs the MI355X" in (r.stderr + r.stdout)


def test_autotune_alias_guard():
    """ops._aliases decides whether a conv may be tile-tuned in place (the tuner re-launches into the caller's output): outputs that
    overlap an input / aux operand must be detected, disjoint channel or row slices of one buffer must not."""
    import torch
    from fgt_amd import ops
    buf = torch.zeros(6, 5, 384)
    assert not ops._aliases(buf[..., :128], None, buf[..., 128:]) and ops._aliases(buf[..., :128], None, buf[..., :128])
    assert ops._aliases(buf[..., :200], None, None, buf[..., 128:])
    rows = torch.zeros(100, 64)
    assert not ops._aliases(rows[:50], None, rows[50:]) and ops._aliases(rows[:60], None, rows[50:])
    assert not ops._aliases(torch.zeros(3, 4), None, rows) and not ops._aliases(None, None, rows)
    # views whose outer strides are NOT multiples of the pixel stride (padded rows): the channel-window shortcut does not apply,
    # the guard must answer "overlaps"  — here element ranges interleave although the channel windows look disjoint
    st = torch.zeros(4 * 1000)
    a = st.as_strided((4, 3, 64), (1000, 200, 1), 0)           # frame stride 1000 = 5 * 200: regular
    b = st.as_strided((4, 3, 64), (1000, 200, 1), 64)
    assert not ops._aliases(a, None, b)
    c = st.as_strided((3, 3, 64), (1100, 200, 1), 0)           # frame stride 1100: not a multiple of 200
    d = st.as_strided((3, 3, 64), (1100, 200, 1), 64)
    assert ops._aliases(c, None, d)


def test_tile_tables_agree_with_the_header():
    """Every tile name the autotuner may pick maps to a code the header defines (FGT_TILE_* [+ 100 wide, + 200 tap-reusing]); the tap-reusing
    kernel's candidates are its own family only — its accumulation order differs, so tuning must never cross families."""
    import re
    from fgt_amd import _lib, ops
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "fgt_hip.h")).read()
    base = {int(v) for v in re.findall(r"#define FGT_TILE_\w+ (\d+)\b", hdr)}
    for name in ops.TILE_CANDIDATES:
        code = _lib.TILE[name]
        assert code < 200 and (code % 100) in base, name
    for name in ops.TAPS_CANDIDATES:
        code = _lib.TILE[name]
        assert 200 <= code < 300 and (code - 200) in base and name.endswith("t"), name
    assert not set(ops.TILE_CANDIDATES) & set(ops.TAPS_CANDIDATES)


def test_entry_build_runs_here():
    """__graft_entry__.build(): what the driver runs on the CPU box every round (compile for gfx950, import, ABI version)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "built" in r.stdout
