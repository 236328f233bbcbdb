#!/usr/bin/env python3
"""Census of the kernel instantiations in the built liboeh_hip.so (CPU; code-object metadata) with the reason each group exists:
    python tools/instantiations.py > profiles/rNN_instantiations.txt
A group = one kernel template x the template arguments that are NOT just (head dim, storage dtype, rows per wave); its count is the
number of (head dim, dtype, NT / MQ) combinations built.  VERDICT r3 next #6 asked for >= 100 fewer instantiations "or a stated
reason for each that stays": every group below is reachable from `pick_variant` / the launchers for a problem some caller of the
reference's modules can pose (dtype x head dim x row length x mask kind x softmax kind x quantisers), and none of them is a
tuning duplicate - two groups never compute the same problem class."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True).stdout


def targs(name):
    return [int(x) if x.lstrip("-").isdigit() else x for x in re.findall(r"L[ib](-?\d+)E", name.replace("Ln", "L-"))]


groups = collections.OrderedDict()
REASON = {
    "flash": {
        (0, 0, 0, 0, 0): "one-pass kernel, plain softmax / softmax_1, masks none | causal: the headline workload (OPT-125m softmax1), long BERT / ViT rows",
        (1, 0, 0, 0, 0): "... with a key-padding vector or a (B,1,Sq,Sk) mask on rows of > 512 keys (PAD: padding row in LDS, trailing padded tiles not streamed)",
        (0, 1, 0, 0, 0): "... with the per-token gate predictor evaluated in the kernel (gated OPT, rows > 128 keys)",
        (1, 1, 0, 0, 0): "... gate predictor + key padding",
        (0, 0, 0, 1, 0): "two-pass clipped softmax on rows of > 512 keys (d = 128: from 384 keys) - what otherwise runs the any-shape kernel (~100x slower)",
        (1, 0, 0, 1, 0): "... with key padding",
        (0, 0, 0, 2, 0): "two-pass fused INT8 chain on rows of > 512 keys",
        (1, 0, 0, 2, 0): "... with a 0 / finfo.min key-padding vector",
        (0, 0, 1, 0, 0): "fp32 storage read in place (fp16 operand pairs): the reference's validate precision, plain softmax(_1)",
        (1, 0, 1, 0, 0): "... fp32 storage + key padding / full mask on long rows",
        (0, 0, 1, 1, 0): "fp32 storage, two-pass clipped softmax on long rows", (1, 0, 1, 1, 0): "... + key padding",
        (0, 0, 1, 2, 0): "fp32 storage, two-pass INT8 chain on long rows", (1, 0, 1, 2, 0): "... + key padding",
        (0, 0, 0, 0, 1): "O32 sibling (round 4): 16-bit storage, output from the fp32 accumulators - how tests / smoke / bench measure the 1e-3 contract on the loop that ships (d = 64 only)",
    },
}
for line in out.splitlines():
    name = line.split()[0]
    m = re.match(r"_ZN3oeh(?:12_GLOBAL__N_1)?\d+(\w+?)(?:I|E)", name)
    fam = m.group(1) if m else name
    a = targs(name)
    key = (fam,)
    if fam == "oeh_attn_flash_kernel":       # <D, IN, MQ, PAD, GATE, SRC32, TP, O32>
        a = a + [0] * (8 - len(a))
        key = (fam, "PAD=%d GATE=%d SRC32=%d TP=%d O32=%d" % tuple(a[3:8]), REASON["flash"].get(tuple(a[3:8]), ""))
    elif fam == "oeh_attn_fast_kernel":      # <NT, D, IN, CLIP, GATE, FQ, SRC32, O32>
        a = a + [0] * (8 - len(a))
        why = {0: "plain / clipped softmax with the whole score row in registers (Sk <= 512): clipped softmax (cfg3), rows <= 128 keys (BERT-base, cfg2 / cfg5), vanilla + key padding",
               1: "the fused INT8 chain on the quantiser grid (cfg4: OPT + --quantize)", 2: "the INT8 chain in the reference's literal op order (key padding with arbitrary additive values)",
               3: "the INT8 grid chain with a 0 / finfo.min key-padding vector (quantised BERT, padded OPT batches)"}[a[5]]
        if a[4]:
            why = "... with the per-token gate predictor in the kernel (gated BERT, cfg5)"
        if a[7]:
            why = "O32 sibling (round 4): output from the fp32 accumulators, for the 1e-3 contract checks (d = 64)"
        key = (fam, "CLIP=%d GATE=%d FQ=%d SRC32=%d O32=%d" % (a[3], a[4], a[5], a[6], a[7]), why + ("; fp32 storage (operand pairs)" if a[6] else ""))
    elif fam == "oeh_attn_mfma_kernel":      # <NT, D, IN, FQ>
        key = (fam, "FQ=%d" % (a[3] if len(a) > 3 else 0), "general kernel: (B,1,Sq,Sk) masks, true score division, gamma > 0, any subset of the quantisers, the test-only index dumps; Sk <= 512")
    elif fam == "oeh_attn_i8_kernel":        # <NT, OUT, DUMP, CQ2, PAD>
        key = (fam, "DUMP=%d CQ2=%d PAD=%d" % tuple(a[2:5]), "INT8 storage on v_mfma_i32_16x16x64_i8 (NT in {8, 16, 32} x output f16 / bf16 / f32 / int8 indices); DUMP: index dumps for the reference-capture tests; CQ2: q grid with zero point 0; PAD: key padding")
    elif fam == "oeh_gemm_kernel":           # <AM, MI, NJ>
        form = {0: "fp16 activations", 1: "fp16 operand pairs of fp32 activations", 2: "fp32 activations, split at fragment-read time",
                3: "int8 x int8 on v_mfma_i32_16x16x64_i8 (out_proj on the context quantiser's indices)"}.get(a[0], "?")
        key = (fam, "AM=%d" % a[0], "projection GEMM with the quantisers in its epilogue (oeh_proj_quant_i8): " + form + "; tiles 128 x 288 and 64 x 192")
    elif fam == "oeh_attn_small_kernel":
        key = (fam, "", "STanHop Association: one wave per (batch, head), exact fp32 products; d in {16, 32, 64} x dtype x rows <= 32 / 64")
    groups[key] = groups.get(key, 0) + 1
total = 0
for key, n in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1:] )):
    total += n
    print(f"{n:4d}  {key[0]:32s} {key[1] if len(key) > 1 else '':44s} {key[2] if len(key) > 2 else ''}")
print(f"{total:4d}  kernels in liboeh_hip.so")
