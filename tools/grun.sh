#!/bin/bash
# build everything here (hipcc cross-compiles without a GPU), then run the given command on an MI355X box
# usage: tools/grun.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python -m tagdust_amd.build > /dev/null
make -s -C oracle libtd_oracle.so
# the GPU box gets no .git: leave the commit (and whether the tree differs from it) for the profile summaries
echo "$(git rev-parse --short=12 HEAD)$(git diff --quiet HEAD -- . || echo +dirty)" > .build_head
exec /usr/local/graft/bin/gpurun "$@"
