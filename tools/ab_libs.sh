#!/bin/bash
# A/B of library builds (simd-raytracer_amd/build_var/librtk_<tag>.so): config-2 frame at world 1 and the config 3 / 4 / 5 shapes.
# usage: tools/ab_libs.sh tag...   (on the GPU box)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"
  L=$GRAFT_REPO_ROOT/simd-raytracer_amd/build_var/librtk_$v.so
  [ -n "$AB_SKIP_CFG2" ] || RTK_LIB_OVERRIDE=$L TC_WORLDS=1 TC_MODES=0 python tools/rank_times.py 2>&1 | grep "world 1"
  RTK_LIB_OVERRIDE=$L TC_MODES="0" python tools/time_configs.py cfg3 cfg4 ${AB_CASES:-} 2>&1 | grep -E "cfg"
done
