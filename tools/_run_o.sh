set -e
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh "" "--tune 3=384" "--tune 3=512" "--tune 3=192"
