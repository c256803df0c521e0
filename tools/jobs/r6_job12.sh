#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
for v in "" "--graph"; do for t in "" "--tail"; do
  python3 bench.py --config detect $v $t --steps 12 --warmup 3 > $out/r6_l_detect_tmp.json 2> $out/r6_l_detect_tmp.err
  python3 -c "
import json; d=json.load(open('$out/r6_l_detect_tmp.json')); c=d['config']; print('detect [$v] [$t]', d['value'], d['ms_per_step'], c.get('hip_graph'), c.get('hip_graph_error'), c.get('rle_masks_encoded'), d['roofline']['frac'])" || tail -3 $out/r6_l_detect_tmp.err
  cp $out/r6_l_detect_tmp.json "$out/r6_l_detect$(echo $v$t | tr -d ' -').json"
done; done
