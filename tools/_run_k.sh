set -e
R=$GRAFT_REPO_ROOT; cd $R
bash tools/ab_bench.sh "" "--tune 25=0"
bash tools/ab_bench.sh "" "--tune 25=0"
