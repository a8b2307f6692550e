#!/bin/bash
# even-M variants one at a time on top of the default policy (round 3)
D="w1,w3,w5,w7,w9,w11,w13,p7,p9,p11,q7,q9,q11,q13,16x4,16x8,16x16"
bash tools/policy_sweep.sh C2 "$D" "$D,w12" "$D,p8" "$D,p10" "$D,w10" "$D,w8" "$D,w6" "$D,w2,w4" "$D"
