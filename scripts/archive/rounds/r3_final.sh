#!/bin/bash
# round-3 evidence run on the GPU box: kernel-trace stats + FETCH/WRITE_SIZE passes + traffic summaries (with plan fingerprints)
# for the bench workloads, the default bench line, the f2 scale test.  Outputs under gpurun_out/prof_r03_*.
cd $GRAFT_REPO_ROOT
[ -x scripts/micro/stream_patterns ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_patterns.hip -o scripts/micro/stream_patterns
for spec in ${@:-laplacian4096:f64 powerlaw8000000:f64 webbase:f64 scircuit:f64 nlpkkt160:f32 nlpkkt160:f64 lap3d256:f64}; do
  wl=${spec%%:*}; dt=${spec#*:}
  echo "== profile $wl $dt"
  timeout -k 10 1000 bash scripts/profile_traffic.sh r03_${wl}_${dt} $wl $dt 2>&1 | tail -4
done
