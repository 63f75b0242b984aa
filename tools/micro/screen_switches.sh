# one-repetition screen of run-time tunables on a workload (each line: one bench run of 20 steps)
#   bash tools/micro/screen_switches.sh                       (the built-in list)
#   SPECS="SMCP_FAM_MINRHS:1 2;SMCP_DOWN_W:1 0" WL=synth50k bash tools/micro/screen_switches.sh
export REPS=1
DEFAULT="SMCP_GRAM_MINCHUNK:512 1024 4096;SMCP_N16_THR_LEAF:128 64 256;SMCP_DIAG_THREADS:512 1024 256;SMCP_POTRF_THREADS:1024 512;SMCP_UPDP_PAD:0 8 16;SMCP_LG_EARLY:1 0;SMCP_SCALING_OVERLAP:1 0;SMCP_AUX_PRIO1:1 0;SMCP_GRAM_NW:16 8;SMCP_FTHR_CHOL:128 64 256;SMCP_FTHR_PINV:256 128 512;SMCP_FTHR_YAA:64 128;SMCP_FTHR_YAA_MID:256 128;SMCP_FAMT_G:0 1 2 3;SMCP_FAM_MINRHS:1 2;SMCP_DOWN_W:1 0"
IFS=';' read -ra LIST <<< "${SPECS:-$DEFAULT}"
for spec in "${LIST[@]}"; do
  sw=${spec%%:*}; vals=${spec#*:}
  bash tools/ab_switch.sh $sw "$vals" ${WL:-synth50k} 20 | grep -v "^    "
done
