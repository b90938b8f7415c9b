#!/bin/bash
# rocprofv3 PMC pass over one forward (developer tool); usage: tools/pmc.sh <outname> <counters...>
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -o $name -- python $GRAFT_REPO_ROOT/tools/profile_layers.py cfg2 > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/$name | head
