#!/bin/bash
# (tuning build) strip length of the conv1_1-in-the-loader kernel: kernel stats of one bench run each, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_fuse1e.txt
: > $OUT
for S in 1 4 2 8; do
  export WCT_FUSE1_STRIP=$S
  bash tools/gpu_prof_stats.sh r04_fuse1e_$S > /dev/null 2>&1
  echo "== strip $S" >> $OUT
  grep "conv3x3" gpurun_out/r04_fuse1e_${S}_kernel_stats.csv | cut -c1-130 >> $OUT
done
export WCT_FUSE1_STRIP=1
export WCT_FUSE_CONV1=0
bash tools/gpu_prof_stats.sh r04_fuse1e_off > /dev/null 2>&1
echo "== unfused" >> $OUT
grep "conv" gpurun_out/r04_fuse1e_off_kernel_stats.csv | cut -c1-130 >> $OUT
cat $OUT
