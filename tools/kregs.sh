#!/bin/bash
# developer: register / scratch use of the kernels in one HIP source (hipcc --save-temps, reads the .s)
# usage: tools/kregs.sh fast_limo_amd/csrc/hip/flimo_kernels.hip [name filter]
set -e
src=$1; filt=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value \
  -c "$src" -o $tmp/k.o --save-temps=obj
s=$(ls $tmp/*gfx950*.s | head -1)
awk '/^[[:space:]]*\.amdhsa_kernel /{k=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_private_segment_fixed_size/{p=$2} /\.amdhsa_accum_offset/{a=$2} /^[[:space:]]*\.end_amdhsa_kernel/{print v, a, p, k}' $s \
  | while read v a p k; do echo "vgpr=$v accum_off=$a scratch=$p $(echo $k | c++filt | cut -c1-110)"; done | grep -E "$filt"
rm -rf $tmp
