#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
JD_VERBOSE=1 python tools/pf_ab.py c2 1 two= 2>&1 | grep -E "k_search|cut short|variant" | tail -40
