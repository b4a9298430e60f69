#!/bin/bash
run() { echo "== $*"; env "$@" python bench/dbg_sw.py 2>&1 | grep -E "err " | cut -c1-95; }
run A=1
run CTD_JIT_EXTRA=-DCTD_NO_FOLD
run CTD_JIT_KEEP_WAVES=1 CTD_JIT_EXTRA=-DCTD_NO_FOLD
run CTD_BLOCK=64
run CTD_JIT_EXTRA=-O1
run CTD_JIT_KEEP_WAVES=1 CTD_JIT_EXTRA=-O1
