#!/bin/bash
# wide-form dispatch thresholds (gram_tall_diag_applies) from the environment of a -DPMT_TUNING build of gram_tall.hip: [MAXCOLS=..] tools/diag_ab.sh shapes..
L=$PWD/parametron.jl_amd/lib_variants/tun.so
f() { sed -e 's/ lda=[0-9]*//' -e 's/([0-9. of]*)//' -e 's/| Q.*//' -e 's/_kernel//g' | cut -c1-220; }
echo "[shipped thresholds]"; PMT_LIB_PATH=$L python tools/tall_probe.py "$@" 2>&1 | grep "^r=" | f
echo "[diagonal tiles fused up to ${MAXCOLS:-4096} columns]"; PMT_TALL_DIAG_MAXCOLS=${MAXCOLS:-4096} PMT_LIB_PATH=$L python tools/tall_probe.py "$@" 2>&1 | grep "^r=" | f
