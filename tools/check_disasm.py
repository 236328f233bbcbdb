#!/usr/bin/env python3
"""Build-time checks on the DISASSEMBLY of the built library (CPU; the build container has no GPU) - the two places where the
kernels rely on something the compiler does not model:

  1. M0.  The LDS-DMA requests are inline asm that writes M0 (`s_mov_b32 m0, sN ; s_nop 0 ; global_load_lds_dwordx4`) and declares
     it clobbered instead of saving / restoring it (oeh_common.h: glds16*_m0).  That is sound only while the COMPILER never keeps a
     value of its own in M0 across such a statement: any compiler-emitted M0 use - s_set_gpr_idx_on / s_movrel* / v_movrel*
     (dynamic register indexing), v_readlane / v_writelane / v_interp / ds_gws* / s_sendmsg* with an m0 operand, or an `m0`
     operand anywhere outside the request pattern - fails the check.
  2. trans -> VALU.  On gfx940 / gfx950 a non-transcendental VALU instruction that reads the result of a transcendental one
     (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos) needs one wait state; LLVM's hazard recogniser inserts it for its own
     instructions but not for inline asm, and the placed tile of oeh_attn_flash.inl issues v_exp_f32 as inline asm.  The check
     fails on a transcendental instruction IMMEDIATELY followed by a reader of its destination.
  3. The headline kernels spill nothing (a spill in the one-pass MQ = 2 kernel doubles the launch: profiles/r06_headline_tile_asm.txt).

    python tools/check_disasm.py            # the built library (or OEH_LIB); exit status 1 on a finding
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f16", "v_log_f16", "v_rcp_f16",
         "v_rsq_f16", "v_sqrt_f16", "v_sin_f16", "v_cos_f16", "v_rcp_iflag_f32", "v_exp_legacy_f32", "v_log_legacy_f32")
M0_FORBIDDEN = ("s_set_gpr_idx", "s_movrel", "v_movrel", "ds_gws", "s_sendmsg", "s_ttrace", "v_interp")
NO_SPILL = ("oeh_attn_flash_kernelILi64ELi0ELi2ELb0ELb0ELb0ELi0ELb0E", "oeh_attn_flash_kernelILi64ELi1ELi2ELb0ELb0ELb0ELi0ELb0E")


def regs_of(tok: str):
    """VGPR numbers named by an operand token: v12, v[12:15], -v3, |v3| ..."""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_code_object(co: str, findings: list, stats: dict):
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co], capture_output=True, text=True).stdout
    check_text(dis, findings, stats)


def check_text(dis: str, findings: list, stats: dict):
    """The rules on one disassembly listing (llvm-objdump -d text); tests/test_abi.py feeds it synthetic listings to show each rule fires."""
    kernel = "?"
    prev = None  # (mnemonic, operands)
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]* ?<(\S+)>:", line)
        if m:
            kernel, prev = m.group(1), None
            continue
        t = line.split("//")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        parts = t.split(None, 1)
        mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        stats["instructions"] = stats.get("instructions", 0) + 1
        # ---- 1. M0
        if any(mn.startswith(f) for f in M0_FORBIDDEN):
            if not (mn.startswith("s_sendmsg") and "MSG_DEALLOC_VGPRS" in ops):  # (an immediate message: no M0 payload)
                findings.append(f"{kernel}: compiler-emitted M0 user `{t}`")
        if re.search(r"\bm0\b", ops):
            ok = (mn == "s_mov_b32" and (ops.startswith("m0,") or ops.endswith(", m0")))  # request: m0 <- sN; keep-forms: sN <- m0 / m0 <- sN
            if not ok:
                findings.append(f"{kernel}: M0 operand outside the LDS-DMA request pattern `{t}`")
            else:
                stats["m0_writes"] = stats.get("m0_writes", 0) + 1
        # ---- 2. trans -> reader
        if prev is not None and any(prev[0].startswith(x) for x in TRANS):
            dst = regs_of(prev[1].split(",")[0])
            srcs = ops.split(",", 1)[1] if "," in ops else ""
            is_trans = any(mn.startswith(x) for x in TRANS)
            if mn.startswith("v_") and not is_trans and dst & regs_of(srcs):
                findings.append(f"{kernel}: `{prev[0]} {prev[1]}` immediately followed by its reader `{t}` (trans -> VALU needs one wait state)")
            stats["trans"] = stats.get("trans", 0) + 1
        prev = (mn, ops)
    return


def main():
    tmp = tempfile.mkdtemp()
    findings, stats = [], {}
    try:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(os.environ.get("OEH_LIB") or os.path.join(ROOT, "outeffhop_amd", "lib", "liboeh_hip.so"), lib)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", lib], check=True, capture_output=True, cwd=tmp)
        cos = sorted(glob.glob(lib + ".*gfx950"))
        if not cos:
            print("no gfx950 code object in the library")
            return 1
        for co in cos:
            check_code_object(co, findings, stats)
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in txt.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                sp = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
                if sp and any(k in name for k in NO_SPILL):
                    findings.append(f"{name}: {sp} spilled VGPRs in a kernel that must not spill")
                stats["kernels"] = stats.get("kernels", 0) + 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"checked {stats.get('kernels', 0)} kernels, {stats.get('instructions', 0)} instructions: {stats.get('m0_writes', 0)} M0 moves (all in the request "
          f"pattern), {stats.get('trans', 0)} transcendental instructions; findings: {len(findings)}")
    for f in findings[:50]:
        print("  " + f)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
