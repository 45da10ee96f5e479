#!/bin/bash
# Round 6 evidence passes on the GPU box: kernel trace + FETCH / WRITE counter passes + traffic summary (scripts/profile_traffic.sh) for the headline and the BASELINE stand-ins,
# then the driver's own bench command.  usage (through gpurun): scripts/profile_r6.sh [workload:dtype ...]
export TILESPMV_ROUND_TAG="round 6"
cd $GRAFT_REPO_ROOT
make -C scripts/micro stream_patterns > /dev/null 2>&1 || (cd scripts/micro && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 stream_patterns.hip -o stream_patterns)
for spec in ${@:-laplacian4096:f64 nlpkkt160:f32 scircuit:f64 webbase:f64 powerlaw8000000:f64 fem3_68:f64 band40_2000000:f64}; do
  wl=${spec%%:*}; dt=${spec#*:}
  echo "== $wl $dt"
  unset TILESPMV_X_PANEL_MERGE TILESPMV_X_SLICE_PASSES
  scripts/profile_traffic.sh r06_${wl}_${dt} $wl $dt 2>&1 | tail -6
done
