#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
for d in 0 1 2 7; do echo "## BIG_DBG=$d (1: no DMA in the loop, 2: no MFMA, 4: no barrier)"; BIG_DBG=$d python tools/exp/big_gemm/run.py 2>&1 | grep "^M="; done
python tools/exp/big_gemm/run.py 2>&1 | grep "^M="
