#!/bin/bash
# A/B two builds of libdvsg_amd.so on the same GPU box, layer by layer: tools/ab_conv.sh <base.so> [conv_bench flags]
# (run through gpurun; build the base from another commit with `git worktree add`).  Prints base, new, base, new.
set -euo pipefail
BASE=$1; shift
for rep in 1 2; do
  echo "== base"; DVSG_AMD_LIB=$BASE python tools/conv_bench.py --variants 0 "$@"
  echo "== new";  python tools/conv_bench.py --variants 0 "$@"
done
