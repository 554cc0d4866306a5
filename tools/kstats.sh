#!/bin/bash
# rocprofv3 kernel-trace summary of a command:  bash tools/kstats.sh <out.txt> -- <cmd...>   (run from the repo root on the GPU box)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift 2
D=$(mktemp -d /tmp/kstats.XXXXXX)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $D -- "$@" > $D/stdout.txt 2> $D/stderr.txt
cd $R
python tools/rocpd_summary.py $D $OUT > /dev/null 2>&1
rm -rf $D
