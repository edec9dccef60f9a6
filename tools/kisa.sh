#!/bin/bash
# usage: tools/kisa.sh <file.hip> <mangled-name-substring> <out.s>  -- ISA of one kernel (hipcc -save-temps), gfx950
set -e
src=$(realpath $1); d=$(dirname $src); tmp=/tmp/kisa_$$; mkdir -p $tmp
(cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$d/../../include -save-temps=obj -c $(basename $src) -o $tmp/x.o ${@:4})
S=$(ls $tmp/*gfx950.s)
a=$(grep -n "^_Z.*$2.*:" $S | head -1 | cut -d: -f1)
b=$(awk -v a=$a 'NR>a && /s_endpgm/ {print NR; exit}' $S)
sed -n "${a},${b}p" $S > $3
rm -rf $tmp
wc -l $3
