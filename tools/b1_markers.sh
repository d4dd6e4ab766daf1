#!/bin/bash
# Per-stage GPU time of the batch-1 path (run ON THE GPU BOX): tools/b1_markers.sh TAG H W [precision]  ->  gpurun_out/TAG_b1_HxW_ranges.csv
set -uo pipefail
TAG=$1; H=$2; W=$3; PREC=${4:-f32}
export TMPDIR=/tmp
O=gpurun_out/prof_b1_${H}x${W}
rm -rf $O
DVSG_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O -- python3 tools/b1_profile.py $H $W $PREC > $O.log 2>&1
python3 tools/marker_summary.py $O gpurun_out/${TAG}_b1_${H}x${W}_ranges.csv > /dev/null 2>&1
cat gpurun_out/${TAG}_b1_${H}x${W}_ranges.csv
python3 tools/stats_sum.py $O 14
