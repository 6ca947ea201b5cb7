cd $GRAFT_REPO_ROOT
bash scratch/ab.sh "nbin32 nbin64" | grep -v "nodes "
