#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for sl in 152 160 168; do for pc in 12288 24576; do echo "== slots $sl piece $pc"; JD_VERBOSE=1 JD_PIPE_PIECE=$pc PIPE_AB_STEPS=30 timeout 600 python tools/pipe_ab.py $sl 2>&1 | grep "resident pipeline\|^pipeline:"; done; done
