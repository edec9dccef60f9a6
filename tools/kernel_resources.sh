#!/bin/bash
# usage: res.sh file.hip pattern  -> prints resource usage of kernels whose mangled name matches pattern
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/res.o 2>&1 | grep -E "Function Name|VGPRs:|SGPRs Spill|VGPRs Spill|Occupancy|LDS Size" | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - - | grep -E "$2" | sed 's/Function Name: //' | awk '{printf "%s\n", $0}' | cut -c1-230
