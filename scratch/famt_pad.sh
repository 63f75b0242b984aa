# speed of k_fam_terms over fresh processes for several paddings of the per-right-hand-side stride of the exchange buffer
# (interleaved, so that a drift of the box over the minutes of the run does not look like an effect of the padding)
cd $GRAFT_REPO_ROOT
for pad in 4128 0 16416 0 544 0 4128 0 32 0 16416 0 544 0; do
  t=$(SMCP_UPDP_PAD=$pad TRIALS=1 timeout -k 10 120 python3 scratch/famt_realloc.py 2>/dev/null | grep "trial 0" | sed 's/.*k_fam_terms \([0-9.]*\) ms.*/\1/')
  echo "pad $pad: $t"
done
