#!/bin/bash
# round 6: ordered kernel timelines of one 128-image get_codes and one 128-image decode_code (per-layer durations), then SQ counters of the halo
# convs inside get_codes
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
for what in encode decode; do
  cd /tmp; rm -rf /tmp/tl_$what
  RQ_WHAT=$what timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$what -o tl -- python $R/scripts/decode_timeline.py run > /dev/null 2>&1
  cd $R; python scripts/decode_timeline.py report /tmp/tl_$what > gpurun_out/r06_${what}_timeline_b128.txt; tail -3 gpurun_out/r06_${what}_timeline_b128.txt
done
RQ_TAG=r06_encode_halo RQ_PMC_CMD="python $R/scripts/encode_trace.py" RQ_PMC_FILTER=conv3x3_halo bash scripts/gpu.sh sqpmc > /dev/null 2>&1; cat gpurun_out/r06_encode_halo_sqpmc.txt | head -40
