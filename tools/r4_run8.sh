#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
python tools/phase_trace.py --two 2>&1 | tail -14
