cd $GRAFT_REPO_ROOT
bash scratch/ab_rh.sh "scf_k0 scf_early"
bash scratch/ab.sh "scf_k0 scf_early scf_k0 scf_early" | grep -v "nodes "
