#!/bin/bash
# round-5 evidence run on the GPU box: kernel-trace stats + FETCH/WRITE_SIZE passes + traffic summaries (with plan fingerprints) for the bench workloads (now with the FEM class),
# outputs under gpurun_out/prof_r05_*; scripts/collect_profiles.sh r05 copies the judged summaries into profiles/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp TILESPMV_ROUND_TAG="round 5"
[ -x scripts/micro/stream_patterns ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_patterns.hip -o scripts/micro/stream_patterns
for spec in ${@:-laplacian4096:f64 fem3_68:f64 fem6_46:f64 fem3s64_68:f64 powerlaw8000000:f64 webbase:f64 scircuit:f64 nlpkkt160:f32 nlpkkt160:f64 lap3d256:f64 band40_2000000:f64 bandrand4x3_2000000:f64 uniform8_4000000:f64 uniform8_8000000:f64}; do
  wl=${spec%%:*}; dt=${spec#*:}
  echo "== profile $wl $dt"
  timeout -k 10 1000 bash scripts/profile_traffic.sh r05_${wl}_${dt} $wl $dt 2>&1 | tail -4
done
