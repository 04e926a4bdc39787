#!/bin/bash
# Code size, VGPR/SGPR counts and scratch of every sim_kernel build in madsim_amd/csrc/sim_kernel.o (no GPU needed).
set -e
B=/opt/rocm/lib/llvm/bin; O=${1:-$(dirname "$0")/../madsim_amd/csrc/sim_kernel.o}
objcopy -O binary --only-section=.hip_fatbin $O /tmp/madsim_kernel_meta.fat
T=$($B/clang-offload-bundler --list --type=o --input=/tmp/madsim_kernel_meta.fat | grep gfx950)
$B/clang-offload-bundler --unbundle --type=o --input=/tmp/madsim_kernel_meta.fat --targets=$T --output=/tmp/madsim_kernel_meta.co
$B/llvm-readelf -sW /tmp/madsim_kernel_meta.co | awk '$4=="FUNC"{print $8, $3}' | sort -u > /tmp/madsim_kernel_sizes.txt
$B/llvm-readelf --notes /tmp/madsim_kernel_meta.co | awk '
  /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {print name, "vgpr", v, "sgpr", s, "scratch", scr, "spills", $2}' > /tmp/madsim_kernel_regs.txt
join <(sort /tmp/madsim_kernel_sizes.txt) <(sort /tmp/madsim_kernel_regs.txt) | sed 's/_ZN8madsim_k10sim_kernelINS_7VariantI//; s/EEEEEvNS_7KParamsE//' | awk '{printf "%-40s code %6d B  %s %s %s %s %s %s %s %s\n", $1, $2, $3,$4,$5,$6,$7,$8,$9,$10}' | sort -k3 -n
