#!/bin/bash
# Registers, scratch (spills) and LDS of every kernel of one built object whose name matches a pattern:
#   tools/kernel_regs.sh quantization_amd/csrc/build/product/miq_conv2d_i8.o [pattern]
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co
$L/llvm-readelf --notes $T/dev.co | awk -v pat="${2:-.}" '
  /\.name:/ {name=$2}
  /\.private_segment_fixed_size:/ {scratch=$2}
  /\.sgpr_spill_count:/ {ss=$2}
  /\.vgpr_count:/ {v=$2}
  /\.agpr_count:/ {a=$2}
  /\.vgpr_spill_count:/ {vs=$2; if (name ~ pat) printf "%-110s vgpr %3s agpr %3s scratch %4s B  vgpr spills %s sgpr spills %s\n", substr(name,1,110), v, a, scratch, vs, ss}'
rm -rf $T
