export REPS=1
for spec in "SMCP_GRAM_NW:16 8" "SMCP_FTHR_CHOL:128 64 256" "SMCP_FTHR_PINV:256 128 512" "SMCP_FTHR_YAA:64 128" "SMCP_FTHR_YAA_MID:256 128 512" "SMCP_FAMT_G:0 1 2 3 4"; do
  sw=${spec%%:*}; vals=${spec#*:}
  bash tools/ab_switch.sh $sw "$vals" synth50k 20 | grep -v "^    "
done
