#!/bin/bash
# usage: r5_matrix.sh "ENV=VAL ENV=VAL" "ENV=VAL" ...   each argument = one configuration (own process), 2 alternations
for cfg in "$@"; do
  echo -n "[$cfg] "
  env $cfg python scratch/ab_step.py engine.gather_inputs None None 2 2>&1 | tail -1 | sed 's/engine.gather_inputs=None: //'
done
