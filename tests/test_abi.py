"""CPU-side checks of the C-ABI library: it loads, exports exactly the symbols include/oeh.h declares, validates
arguments without touching a GPU, and the product package refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "oeh.h")


def _declared():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(oeh_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from outeffhop_amd import _lib

    assert _declared() == sorted(_lib.EXPORTS)


def test_library_loads_and_exports_every_symbol():
    from outeffhop_amd import _lib

    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    import shutil

    nm = shutil.which("nm")
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True) if nm else None
    if out is not None and out.returncode == 0:
        exported = {ln.split()[-1] for ln in out.stdout.splitlines() if " T " in ln}
        assert set(_declared()) <= exported
    assert lib.oeh_abi_version() == 6
    assert b"gfx950" in lib.oeh_build_info()
    assert lib.oeh_strerror(-22) == b"invalid argument"


def test_argument_validation_without_gpu():
    from outeffhop_amd import _lib

    lib = _lib.load()
    d = _lib.oeh_attn_desc()
    assert lib.oeh_attn_fwd(None, None, None, None, None, None, None) == -22
    d.B, d.H, d.Sq, d.Sk, d.D, d.dtype = 1, 1, 4, 4, 64, 7
    one = C.c_void_p(16)
    assert lib.oeh_attn_fwd(C.byref(d), one, one, one, one, None, None) == -22  # bad dtype
    d.dtype = 0
    d.softmax_base = 3
    assert lib.oeh_attn_fwd(C.byref(d), one, one, one, one, None, None) == -22  # bad softmax base
    assert lib.oeh_softmax_rows(one, one, 4, 0, 0, 1, 0, 0.0, 1.0, None) == -22
    assert lib.oeh_fake_quant(one, one, None, 4, 0, 0.0, 0.0, 255.0, None) == -22  # scale must be > 0
    assert lib.oeh_fake_quant(one, one, one, 4, 0, 1.0, 0.0, 1023.0, None) == -95  # uint8 dump of a 10-bit grid
    assert lib.oeh_percentile_ema(one, 0, 2, 0.001, 99.999, 0.9, 1, one, one, None) == -22  # empty tensor
    segs = (_lib.oeh_proj_seg * 3)()
    assert lib.oeh_proj_quant_i8(one, 1, one, one, 2, 64, 768, 768, 3, None, 1536, 768, None) == -22  # no segments
    assert lib.oeh_proj_quant_i8(one, 1, one, one, 2, 64, 772, 768, 3, segs, 1544, 772, None) == -95  # K % 32
    assert lib.oeh_proj_quant_i8(one, 1, one, one, 2, 60, 768, 768, 3, segs, 1536, 768, None) == -95  # S % 16
    assert lib.oeh_proj_quant_i8(one, 1, one, one, 2, 64, 768, 768, 3, segs, 768, 768, None) == -22  # pairs need lda >= 2K
    assert lib.oeh_proj_quant_i8(one, 1, one, one, 2, 64, 768, 768, 3, segs, 1536, 768, None) == -22  # a segment with neither out nor y
    assert lib.oeh_percentile_ema(one, 8, 2, 0.001, 100.5, 0.9, 1, one, one, None) == -22  # percent out of range
    assert lib.oeh_percentile_ema(one, 8, 2, 0.001, 99.999, 0.9, 1, C.c_void_p(20), one, None) == -14  # state not 8-byte aligned
    assert lib.oeh_fake_quant_range(one, one, 8, 2, None, 8, 1e-8, None) == -22
    assert lib.oeh_fake_quant_range(one, one, 8, 2, one, 0, 1e-8, None) == -22  # n_bits


def test_variant_selection_host_only():
    import torch
    from outeffhop_amd import ops

    assert ops.attn_variant(16, 12, 512, 512, 64) == "flash16/MQ2/D64/f16"
    assert ops.attn_variant(32, 12, 128, 128, 64) == "fast16/NT8/D64/f16"
    assert ops.attn_variant(2, 6, 197, 197, 64, torch.bfloat16) == "flash16/MQ1/D64/bf16"
    assert ops.attn_variant(1, 1, 8, 100000, 64) == "flash16/MQ1/D64/f16"
    assert ops.attn_variant(16, 12, 512, 512, 64, clip=True) == "fast16/NT32/D64/f16/clip"
    assert ops.attn_variant(2, 6, 197, 197, 64, torch.bfloat16, clip=True) == "fast16/NT16/D64/bf16/clip"
    assert ops.attn_variant(16, 12, 512, 512, 64, fq=True) == "fast16/NT32/D64/f16/fq"
    assert ops.attn_variant(4, 1, 64, 64, 32, torch.float32, fq=True) == "fast16/NT8/D32/f32/fq"
    assert ops.attn_variant(3, 4, 7, 5, 16) == "small/ST2/D16/f16"          # STanHop-sized: one wave per (batch, head)
    assert ops.attn_variant(224, 4, 28, 28, 64, torch.float32) == "small/ST2/D64/f32"
    assert ops.attn_variant(3, 4, 7, 5, 48) == "generic"
    assert ops.attn_variant(1, 1, 8, 100000, 64, torch.float32) == "flash16/MQ1/D64/f32"  # fp32 rows of any length: one-pass kernel
    assert ops.attn_variant(1, 1, 8, 100000, 64, torch.float32, clip=True) == "flash16/MQ1/D64/f32/clip2p"  # two-pass forms: any length
    assert ops.attn_variant(1, 1, 8, 100000, 64, torch.float16, fq=True) == "flash16/MQ1/D64/f16/fq2p"
    assert ops.attn_variant(1, 1, 8, 100000, 64, torch.float32, clip=True, gamma=0.01) is None          # nothing holds a 100000-key row with gamma > 0


def test_no_cpu_fallback():
    import torch
    from outeffhop_amd import _lib, ops

    x = torch.zeros(2, 2, 4, 64, dtype=torch.float16)
    with pytest.raises(_lib.OehError):
        ops.attn_fwd(x, x, x)
    with pytest.raises(_lib.OehError):
        ops.softmax_rows(torch.zeros(2, 3))


def test_fp32_storage_variants():
    """fp32 storage (the reference's validate_* scripts) is read in place by the 16-bit-operand kernels."""
    import torch

    from outeffhop_amd import ops

    assert ops.attn_variant(16, 12, 512, 512, 64, torch.float32) == "flash16/MQ2/D64/f32"
    assert ops.attn_variant(16, 12, 512, 512, 64, torch.float32, clip=True) == "fast16/NT32/D64/f32/clip"
    assert ops.attn_variant(16, 12, 512, 512, 64, torch.float32, fq=True) == "fast16/NT32/D64/f32/fq"
    assert ops.attn_variant(32, 12, 128, 128, 64, torch.float32) == "fast16/NT8/D64/f32"  # (round 3: the full-row fp32 form on short rows, 16.5 vs 20.0 us)
    assert ops.attn_variant(2, 2, 40, 40, 48, torch.float32) == "generic"


def test_debug_hooks_are_inert_unless_enabled():
    """include/oeh_debug.h: the process-global diagnostic hooks change nothing - and say so - unless
    OEH_DEBUG_HOOKS=1 is in the environment when the library is first used."""
    import sys

    code = ("import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); "
            "print(lib.oeh_debug_set_variant(4, 0), lib.oeh_debug_set_stamps(None))")
    from outeffhop_amd import _lib

    env = {k: v for k, v in os.environ.items() if k != "OEH_DEBUG_HOOKS"}
    off = subprocess.run([sys.executable, "-c", code, _lib.LIB_PATH], env=env, capture_output=True, text=True)
    on = subprocess.run([sys.executable, "-c", code, _lib.LIB_PATH], env={**env, "OEH_DEBUG_HOOKS": "1"}, capture_output=True, text=True)
    assert off.stdout.split() == ["-95", "-95"], off.stdout + off.stderr
    assert on.stdout.split() == ["0", "0"], on.stdout + on.stderr
    hdr = open(os.path.join(ROOT, "include", "oeh_debug.h")).read()
    assert "oeh_debug_set_variant" in hdr and "oeh_debug_set_stamps" in hdr
    assert "oeh_debug" not in open(os.path.join(ROOT, "include", "oeh.h")).read()


def test_ctypes_structures_have_the_headers_layout(tmp_path):
    """The ctypes mirrors in outeffhop_amd/_lib.py against include/oeh.h as a C compiler lays it out: size of every structure and
    offset of every field (a field added on one side only, or a type that pads differently, would shift everything behind it)."""
    import shutil

    from outeffhop_amd import _lib

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    # (the (scale, zero_point) pairs q_grid / k_grid / v_grid are an anonymous struct in the header: covered through oeh_attn_desc's offsets)
    structs = {"oeh_fq": _lib.oeh_fq, "oeh_fq_desc": _lib.oeh_fq_desc, "oeh_attn_desc": _lib.oeh_attn_desc, "oeh_proj_seg": _lib.oeh_proj_seg}
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{HDR}"', "int main(void) {"]
    for name, st in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in st._fields_:
            lines.append(f'  printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c99", "-o", str(exe), str(src)], check=True, capture_output=True)
    out = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, st in structs.items():
        assert int(out[name]) == C.sizeof(st), (name, out[name], C.sizeof(st))
        for fname, _ in st._fields_:
            assert int(out[f"{name}.{fname}"]) == getattr(st, fname).offset, (name, fname, out[f"{name}.{fname}"], getattr(st, fname).offset)


def test_a_plain_c_host_links_the_library(tmp_path):
    """include/oeh.h is C99 and the exports have C linkage: a C program compiled with gcc against the header links liboeh_hip.so and
    calls the host-only entry points (and gets EINVAL, not a crash, from the device entry points on NULL arguments)."""
    import shutil

    from outeffhop_amd import _lib

    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("no C compiler or library not built")
    src = tmp_path / "host.c"
    src.write_text(f'''#include <stdio.h>
#include <string.h>
#include "{HDR}"
int main(void) {{
  oeh_attn_desc d;
  memset(&d, 0, sizeof d);
  printf("%d %s %d %d\\n", oeh_abi_version(), oeh_strerror(-95), oeh_attn_fwd(NULL, NULL, NULL, NULL, NULL, NULL, NULL),
         oeh_proj_quant_i8(NULL, 0, NULL, NULL, 1, 16, 32, 64, 1, NULL, 32, 32, NULL));
  return 0;
}}
''')
    exe = tmp_path / "host"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-o", str(exe), str(src), "-L" + libdir, "-loeh_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                    "-Wl,--allow-shlib-undefined"], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "6" and out[-2:] == ["-22", "-22"], out


def test_built_library_disassembly_keeps_the_hand_kept_hazard_rules():
    """the LDS-DMA requests clobber M0 without saving it, and the placed tile issues v_exp_f32 as inline asm - both outside
    what the compiler checks.  tools/check_disasm.py reads the built library's gfx950 disassembly: no compiler-emitted M0 user, no M0
    operand outside the request pattern, no transcendental instruction immediately followed by a reader of its result, no spills in the
    headline kernels."""
    import subprocess
    import sys

    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_disasm.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "findings: 0" in r.stdout


def test_disassembly_rules_fire_on_synthetic_listings():
    """Each rule of tools/check_disasm.py on a listing that breaks it (and not on the patterns the kernels really contain)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("check_disasm", os.path.join(ROOT, "tools", "check_disasm.py"))
    cd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cd)

    def run(body):
        f, st = [], {}
        cd.check_text("0000000000001000 <kernel_a>:\n" + body, f, st)
        return f

    ok = """\ts_mov_b32 m0, s56
\ts_nop 0
\tglobal_load_lds_dwordx4 v106, s[16:17]
\tv_exp_f32_e32 v78, v78
\ts_nop 0
\tv_cvt_pk_f16_f32 v78, v78, v79
\tv_exp_f32_e32 v80, v80
\tv_exp_f32_e32 v81, v81
\tv_mfma_f32_16x16x32_f16 v[86:89], v[132:135], v[74:77], v[86:89]
"""
    assert run(ok) == []
    assert any("trans -> VALU" in x for x in run("\tv_exp_f32_e32 v78, v78\n\tv_cvt_pk_f16_f32 v1, v78, v79\n"))
    assert any("trans -> VALU" in x for x in run("\tv_rcp_f32_e32 v5, v6\n\tv_pk_mul_f32 v[8:9], v[4:5], v[10:11]\n"))   # (register ranges are parsed too)
    assert run("\tv_exp_f32_e32 v78, v78\n\tv_cvt_pk_f16_f32 v78, v79, v80\n") == []      # overwrites, does not read
    assert any("M0 user" in x for x in run("\ts_set_gpr_idx_on s4, gpr_idx(SRC0)\n"))
    assert any("M0 operand" in x for x in run("\tv_readlane_b32 s5, v3, m0\n"))
    assert run("\ts_mov_b32 s7, m0\n\ts_mov_b32 m0, s7\n") == []                           # the save / restore forms of -DOEH_KEEP_M0 builds
