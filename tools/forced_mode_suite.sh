#!/bin/bash
# The `-m gpu` suite under forced library modes: each knob routes EVERY context of every test through a path the
# defaults only take for some problem sizes (round 3 found a stale-weights defect of RIPCG and a capture/upload race this
# way).  Expected: all green (the test of the layout the library picks BY DEFAULT skips itself when the
# environment forces the layout; the at-size oracle comparison runs in every mode).
#   tools/forced_mode_suite.sh [ck|base|det]  (about 4 GPU-minutes per mode; "det": only the bit-reproducible mode's; "ck": only the modes that force the camera-chunk kernels,
#                                          "base": only the others)
cd "$(dirname "$0")/.." || exit 1
# -rf --tb=line: every failure with the assertion that failed, not only the test name
run() { echo "== $*"; env "$@" python3 -m pytest tests -q -m gpu -rf --tb=line 2>&1 | grep "FAILED\|Error\|assert\|passed\|failed"; }
if [ "${1:-all}" = "all" ] || [ "${1:-all}" = "base" ]; then
run POVAR_E0_V1=0
run POVAR_E0_V1=0 POVAR_LPL_PLACE=async POVAR_COLD_Q_ROWS=1
run POVAR_E0_V1=0 POVAR_HOT_ACC=8
run POVAR_E0_V1=0 POVAR_HOT_ACC=8 POVAR_COLD_Q_ROWS=0
run POVAR_E0_V1=0 POVAR_LPL_K0=2 POVAR_E0_WGS=7
run POVAR_E0_V1=0 POVAR_LPL_STRATEGY=range POVAR_HOT_ACC=24 POVAR_LPL_PLACE=async
run POVAR_E0_V1=0 POVAR_LPL_NOGRID=1 POVAR_LONG_SEPARATE=1 POVAR_HOT_ACC=40 POVAR_NO_ERR_MEMO=1
run POVAR_PREPARE_V1=1 POVAR_NO_FUSE=1
run POVAR_NO_GRAPH=1
# the resident power series wherever a context allows it (step 1, LDS-accumulating E0 mode, up to 400 000 observations: POVAR_RES_MAX_OBS; never with peers)
run POVAR_RES=1
run POVAR_RES=1 POVAR_RES_WGS=7 POVAR_NO_GRAPH=1
fi
if [ "${1:-all}" = "det" ] || [ "${1:-all}" = "all" ]; then
# the bit-reproducible mode for every context (e0_ck_det / e0_ck_h_det wherever the layout allows; the tests about what the
# mode pins skip themselves: tests/conftest.py), and its ticket order under stress: few accumulators, many batches / short
# chunks, seven workgroups (dozens of tiles per wavefront: every accumulator's ticket chain runs over many rounds)
run POVAR_DETERMINISTIC=1 POVAR_E0_V1=0
run POVAR_DETERMINISTIC=1 POVAR_E0_V1=0 POVAR_HOT_ACC=24 POVAR_LPL_STRATEGY=range
run POVAR_DETERMINISTIC=1 POVAR_E0_V1=0 POVAR_CK_NB=3 POVAR_CK_HMAX=5 POVAR_NO_GRAPH=1
run POVAR_DETERMINISTIC=1 POVAR_E0_V1=0 POVAR_LPL_K0=2 POVAR_E0_WGS=7
fi
[ "${1:-all}" = "det" ] && exit 0
[ "${1:-all}" = "base" ] && exit 0
run POVAR_E0_V1=0 POVAR_E0_CK=1 POVAR_LPL_PLACE=sync
# the camera-chunk kernels (e0_ck, and e0_ck_h in every step-2 context) under stress: few accumulators (most chunks write their own record), many batches / short chunks,
# seven workgroups (dozens of tiles per wavefront), the two-group instantiation, the 12-wavefront one on a layout cut for it
run POVAR_E0_V1=0 POVAR_E0_CK=1 POVAR_HOT_ACC=24 POVAR_LPL_STRATEGY=range
run POVAR_E0_V1=0 POVAR_E0_CK=1 POVAR_CK_NB=3 POVAR_CK_HMAX=5 POVAR_LPL_PLACE=async
# step 2's wide stride with a dozen accumulators per workgroup: most cameras of every step-2 context lose their slot (round 6)
run POVAR_E0_V1=0 POVAR_E0_CK=1 POVAR_CKH_STRIDE=2048 POVAR_CKH_ACC_CAP=12 POVAR_LPL_PLACE=sync
run POVAR_E0_V1=0 POVAR_E0_CK=1 POVAR_LPL_K0=2 POVAR_E0_WGS=7
run POVAR_E0_V1=0 POVAR_E0_CK=4 POVAR_CK_NB=4
run POVAR_E0_V1=0 POVAR_E0_CK=3 POVAR_NO_GRAPH=1
