#!/bin/bash
# usage: ab.sh spec... ; runs each spec alternately with main lib and alt lib, 3 rounds
ALT=$PWD/outeffhop_amd/lib/alt/liboeh_hip.so
for r in 1 2 3; do
  echo "== main"; timeout 300 python tools/microbench.py "$@" 2>&1 | grep "us "
  echo "== alt";  OEH_LIB=$ALT timeout 300 python tools/microbench.py "$@" 2>&1 | grep "us "
done
