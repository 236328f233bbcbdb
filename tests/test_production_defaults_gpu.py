"""The `-m gpu` suite runs with the diagnostic hooks of include/oeh_debug.h ENABLED (tests/conftest.py: several tests force a
second kernel variant over the same problem through them).  Production never sets OEH_DEBUG_HOOKS, so a representative part of
the suite - the BASELINE-configuration tests at full size, the reference-golden core and module tests, the INT8 paths - is run
once more in a child process with the hooks off."""
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBSET = ("full_size or reference_golden or golden_core or int8_storage_on or indices_match or calibrate_fix_eval or bert_module or opt_module "
          "or vit_module or stanhop or consecutive_batches")


def test_representative_subset_with_the_debug_hooks_off():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {**os.environ, "OEH_DEBUG_HOOKS": "0"}  # conftest only sets a default
    probe = subprocess.run([sys.executable, "-c", "import os, sys; sys.path.insert(0, %r); from outeffhop_amd import _lib; "
                            "print(_lib.load().oeh_debug_set_variant(0, 0))" % ROOT], env=env, capture_output=True, text=True, cwd=ROOT)
    assert probe.stdout.strip().endswith("-95"), (probe.stdout, probe.stderr[-500:])  # the hooks really are inert in that environment
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_attn_gpu.py", "tests/test_modules_gpu.py", "-k", SUBSET,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT)
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "no tests ran" not in tail, tail
    print("hooks off:", tail.strip().splitlines()[-1])
