#!/bin/bash
# A/B of the working tree's libraries against an earlier commit's, in ONE process per workload (scripts/exp_bench.py LIB=_old): the only
# comparison that separates a code effect from box / instance effects.  (Round 3: a 7-13 % regression of the fp64 y-store path lived for nine
# commits because single runs on different boxes were read as "a slower box".)
#   here:        scripts/ab_prev.sh build <commit>        -> tilespmv_amd/lib/libtilespmv_{f64,f32}_old.so (git worktree under /tmp, removed again)
#   on the box:  scripts/ab_prev.sh run [workload ...]    (through gpurun; the _old libraries travel with the snapshot)
#   afterwards:  scripts/ab_prev.sh clean
# The Python layer of the working tree drives both libraries: the commit must have the same tilespmv_plan_options / info layout.
set -e
cd "$(dirname "$0")/.."
case "$1" in
  build)
    rm -rf /tmp/ab_prev_wt; git worktree add -q /tmp/ab_prev_wt "$2"
    make -C /tmp/ab_prev_wt/tilespmv_amd/csrc -j8 all > /dev/null
    for d in f64 f32; do cp /tmp/ab_prev_wt/tilespmv_amd/lib/libtilespmv_$d.so tilespmv_amd/lib/libtilespmv_${d}_old.so; done
    git worktree remove --force /tmp/ab_prev_wt; ls -la tilespmv_amd/lib ;;
  run)
    shift
    for wl in ${@:-laplacian4096 lap3d256 nlpkkt160 powerlaw8000000 webbase scircuit}; do
      echo "== $wl fp64"; EXP_F64=1 timeout -k 10 500 python scripts/exp_bench.py $wl "Q=1" "LIB=_old,Q=1" "Q=2" "LIB=_old,Q=2" 2>&1 | grep -v amdgpu.ids
      echo "== $wl fp32"; EXP_F32=1 timeout -k 10 500 python scripts/exp_bench.py $wl "Q=1" "LIB=_old,Q=1" "Q=2" "LIB=_old,Q=2" 2>&1 | grep -v amdgpu.ids
    done ;;
  clean) rm -f tilespmv_amd/lib/*_old.so ;;
  *) echo "usage: $0 build <commit> | run [workload ...] | clean"; exit 1 ;;
esac
