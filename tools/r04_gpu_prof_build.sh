#!/bin/bash
# round 4: rocprofv3 kernel stats of the device-side builders (rand_init: svdf_k_init.hip; window data sets: svdf_k_wbuild.hip)
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04build
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_init -o kt -- python tools/init_probe.py > $OUT/init_probe.txt 2> $OUT/init.stderr.log
find $OUT/kt_init -name "*kernel_stats.csv" -exec cp {} $OUT/init_kernel_stats.csv \;
rm -rf $OUT/kt_init
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_wb -o kt -- python tools/wbuild_probe.py > $OUT/wbuild_probe.txt 2> $OUT/wb.stderr.log
find $OUT/kt_wb -name "*kernel_stats.csv" -exec cp {} $OUT/wbuild_kernel_stats.csv \;
rm -rf $OUT/kt_wb
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $OUT/init_probe.txt; head -12 $OUT/init_kernel_stats.csv | cut -c1-160
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $OUT/wbuild_probe.txt; head -24 $OUT/wbuild_kernel_stats.csv | cut -c1-160
