#!/bin/bash
# usage: tools/archive/probes/loop_scratch.sh <kernel.s>  -- per loop (back-edge) of an ISA dump (tools/kisa.sh): lines, scratch / buffer / LDS instructions in the body
f=$1
grep -n "s_cbranch\|s_branch" $f | awk '{print $1,$2,$3}' | while read ln op tgt; do t=$(grep -n "^$tgt:" $f | cut -d: -f1); l=${ln%:}; if [ -n "$t" ] && [ "$t" -lt "$l" ]; then echo "back-edge at $l -> $tgt (line $t): body $((l-t)) lines, scratch $(sed -n "${t},${l}p" $f | grep -c scratch_), buffer $(sed -n "${t},${l}p" $f | grep -c 'buffer_'), ds $(sed -n "${t},${l}p" $f | grep -c 'ds_'), valu $(sed -n "${t},${l}p" $f | grep -c '^\s*v_')"; fi; done
