# one-repetition screen of run-time tunables on the headline (each line: one bench run of 20 steps)
export REPS=1
for spec in ${SPECS:-"SMCP_GRAM_MINCHUNK:512 4096 8192 1024" "SMCP_N16_THR_LEAF:128 64 256" "SMCP_DIAG_THREADS:512 1024 256" "SMCP_POTRF_THREADS:1024 512" "SMCP_UPDP_PAD:0 8 16" "SMCP_LG_EARLY:1 0" "SMCP_SCALING_OVERLAP:1 0" "SMCP_AUX_PRIO1:1 0"}; do
  sw=${spec%%:*}; vals=${spec#*:}
  bash tools/ab_switch.sh $sw "$vals" ${WL:-synth50k} 20 | grep -v "^    "
done
