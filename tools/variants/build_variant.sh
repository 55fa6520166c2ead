#!/bin/bash
# Diagnostic / experiment builds of the library from PATCHED copies of the sources (the shipping sources carry no stamp
# and no experiment branch).  usage: tools/variants/build_variant.sh <patch name without .patch> [out name] [-D flags...]
#   ck_head               the per-camera step of a term at the HEAD of the next e0_ck launch (round 6: built, measured, not shipped;
#                         a record of that experiment: it applies to the sources of commit df549ed, before povar_hip.hip was cut into units --
#                         profiles/r06_experiments.txt D; POVAR_CK_HEAD=0|1, tools/r06_head_probe.py, tools/r06_head_ab.sh)
#   ck_stamps             in-kernel s_memtime stamps of e0_ck's phases (tools/ck_stamps.py) + the timing-only experiment
#                         branches of round 4 (-DPOVAR_CK_EXP_NOATOMIC, -DPOVAR_CK_EXP_NOBWDROWS, -DPOVAR_CK_EXP_NOBWDLDS)
#   ck_touch              one extra load per wavefront that touches G's cache lines ahead of the landmark step (round 6: +5.2 us per term as
#                         the batch's first request, +2.8 behind the first tile's requests; profiles/r06_touch_prefetch_ab.txt, tools/r06_variant_ab.sh)
#   ck_soa16              z and P3 of the camera records in piece-major images of 16-byte pieces (round 6: +1.8 us per term; profiles/r06_soa16_ab.txt)
#   ck_wperm              the tile of a round each wavefront walks from POVAR_CK_WPERM (round 6: tools/r06_wperm_sweep.sh, profiles/r06_wperm_sweep.txt)
#   ck_rows128            z and the static half of the camera records in 128-byte rows (round 6: +1.8 us per term; profiles/r06_rows128_ab.txt)
# (series_res has its own generator: tools/variants/res_stamps.py)
set -e
cd "$(dirname "$0")/../.."
patch=$1; shift
out=${1:-$patch}; [ $# -gt 0 ] && shift
top=build/variants/$out
rm -rf $top; mkdir -p $top/povar_amd
cp -r povar_amd/csrc $top/povar_amd/csrc
cp -r include $top/include
(cd $top && patch -p1 -s < ../../../tools/variants/$patch.patch)
# (the patched copy's own Makefile: five translation units side by side; extra -D flags through CXXFLAGS)
make -s -C $top/povar_amd/csrc OUT=$PWD/build/libpovar_hip_$out.so OBJ_DIR=$PWD/$top/obj \
  CXXFLAGS="-O3 -std=c++17 -fPIC -munsafe-fp-atomics $*" $PWD/build/libpovar_hip_$out.so
echo build/libpovar_hip_$out.so
