#!/bin/bash
# Build libwct_hip.so from another git ref (or from the working tree: ref "WORK") into wct_tf_amd/variants/<name>.so, for same-box
# A-B runs of two BUILDS on the GPU box (tools/gpu_ab_libs.sh).  usage: bash tools/build_variant.sh <name> <git-ref|WORK> [tuning]
set -e
NAME=$1; REF=$2; TUNING=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/wct_tf_amd/variants
T=$(mktemp -d)
if [ "$REF" = WORK ]; then (cd $ROOT && tar -c --exclude='*.so' --exclude='*.o' --exclude=variants wct_tf_amd include) | tar -x -C $T
else (cd $ROOT && git archive $REF wct_tf_amd include) | tar -x -C $T; fi
(cd $T && WCT_BUILD_TUNING=$TUNING python -m wct_tf_amd.build --force > $T/build.log 2>&1) || { tail -20 $T/build.log; exit 1; }
cp $T/wct_tf_amd/libwct_hip.so $ROOT/wct_tf_amd/variants/$NAME.so
rm -rf $T
echo built wct_tf_amd/variants/$NAME.so
